"""Library-GEMM selection for the three MLP GEMMs when they run on hipBLASLt (fc1 forward, fc1 / fc2 input gradients:
the `Mlp.mlp_on_lib = True` A/B configuration; by default they run on this library's eight-wave kernel).

hipBLASLt's default heuristic picks 130-150 us kernels for these shapes at the bench geometry; its own exhaustive search
(PyTorch TunableOp, run once on an MI355X: ``PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py``)
finds 108-119 us ones.  The result file is shipped; this module only switches TunableOp on in *lookup* mode (no tuning at
run time).  A file recorded for another PyTorch / hipBLASLt / GPU is rejected by TunableOp's validators and the default
heuristic stays in charge."""
import os
import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_tuned_gemms(path=TUNED_FILE):
    """Returns True when the tuned selections were loaded."""
    # Off by default since the block GEMMs moved to this library's kernels (nothing in the step that TunableOp covers is
    # hot any more); ACR_TUNED_GEMMS=1 brings the lookup back for the `Mlp.mlp_on_lib` / `Mlp.fused = False` A/B configurations.
    if os.environ.get("ACR_TUNED_GEMMS", "0") != "1" or not torch.cuda.is_available() or not os.path.exists(path):
        return False
    try:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.record_untuned_enable(False)
        return bool(torch.cuda.tunable.read_file(path))
    except Exception:                                      # TunableOp unavailable in this build: keep the default heuristic
        return False


MIOPEN_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_miopen_db():
    """Point MIOpen's *user* find-db at a private copy of the shipped one (recorded on an MI355X for the stem's
    convolution shapes at the bench geometry).  Call before the first convolution.  Without it every fresh machine spends
    ~30 s of the first step benchmarking solvers; with a db recorded for another MIOpen build the files are simply not
    matched (their names carry arch + version) and nothing changes.  No effect when MIOPEN_USER_DB_PATH is already set."""
    import shutil
    import tempfile
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(MIOPEN_DB_DIR):
        return None
    dst = os.path.join(tempfile.gettempdir(), "acr_miopen_db_%d_%s" % (os.getuid(), os.environ.get("LOCAL_RANK", "0")))
    try:
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(MIOPEN_DB_DIR):
            if not os.path.exists(os.path.join(dst, f)):
                shutil.copy(os.path.join(MIOPEN_DB_DIR, f), os.path.join(dst, f))
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
