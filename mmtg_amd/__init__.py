"""mmtg_amd: MI355X-native implementation of MMTG's training + generation hot path.

Public surface mirrors the reference (src/model.py, src/loss.py, src/generate.py,
src/configs.py); the computation is hand-written HIP for gfx950 behind the C ABI
in include/mmtg_hip.h.  Importing this package never imports the oracle.
"""
import os as _os

# Kernel arguments in DEVICE memory (a HIP runtime setting, read when the runtime initialises -- i.e. at the process's first GPU
# call, so it has to be in the environment before that; an explicit setting of the user's wins).  The training step is ~320
# dependent launches whose kernels each begin by fetching their argument block; from host memory that fetch crosses the PCIe link:
# measured on one box, two A/B pairs, 15.46 / 15.49 ms per step without against 14.93 / 14.92 with
# (profiles/r06_v3_train_hip_force_dev_kernarg_ab.txt).  Graph-replayed launches (the decode step) keep theirs on the device already.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .configs import data_config, gpt2_config, make_model_cfgs, model_cfgs  # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so that `import mmtg_amd.synth` stays light
    if name in ("MMTG", "GPT2_Decoder", "MultiModalEncoder", "InnerModalAttentionLayer", "MultiModalAttentionLayer"):
        from . import model
        return getattr(model, name)
    if name == "MyLoss":
        from .loss import MyLoss
        return MyLoss
    if name in ("pack_token_table", "load_token_table"):
        from . import model
        return getattr(model, name)
    if name in ("sample_sequence", "top_k_top_p_filtering", "generate_samples", "postprocess_tokens"):
        from . import generate
        return getattr(generate, name)
    if name in ("MyDataset", "BinaryDataset", "DeviceLoader", "pack_binary"):
        from . import data
        return getattr(data, name)
    if name in ("MMTGTrainer",):
        from . import trainer
        return getattr(trainer, name)
    raise AttributeError(name)
