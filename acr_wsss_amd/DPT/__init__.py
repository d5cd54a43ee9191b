"""Mirror of the reference's ``DPT`` package for the ACR path: ``from acr_wsss_amd.DPT.ACR import ACR``."""
