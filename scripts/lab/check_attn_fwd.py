"""GPU lab check: fp32 attention forward (o, lse2) against fp64 torch at awkward T, repeated to expose races."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
for (B, T, H) in ((2, 145, 12), (2, 129, 12), (1, 160, 12), (3, 97, 5), (1, 785, 12)):
    g = torch.Generator().manual_seed(T)
    qkv = (1.5 * torch.randn(B, T, 3 * H * 64, generator=g)).to(dev)
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    S = (q @ k.transpose(-2, -1)) * 64 ** -0.5
    lse_ref = torch.logsumexp(S, -1) / 0.6931471805599453
    o_ref = (S.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
    for rep in range(4):
        o, _ = ops.attention_core(qkv.clone().requires_grad_(True), H, None, 0, None)
        lse2 = o.grad_fn.saved_tensors[2]
        el = (lse2.double() - lse_ref).abs()
        eo = (o.double() - o_ref).abs().amax(-1)
        bad = (el > 1e-4).nonzero()
        print("B%d T%d H%d rep%d: max lse err %.2e, max o err %.2e, bad rows %d %s" % (B, T, H, rep, el.max(), eo.max(), bad.shape[0],
              bad[:5].tolist() if bad.shape[0] else ""))
