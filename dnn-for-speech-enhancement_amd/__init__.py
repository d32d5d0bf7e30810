"""MI355X-native replacement of the reference's frame-wise DNN trainer hot path
(`BP_GPU` in yongxuUSTC/DNN-for-speech-enhancement).  The product is the gfx950 HIP library
`libbp_hip.so` behind the C ABI in include/bp_c_api.h; this package is its Python host mirror.
(The directory name contains '-', import it through `dnnse_amd.py` at the repo root.)"""
from .bp_gpu import (BP_GPU, BPError, BPConfig, load_library, LIB_PATH, ABI_SYMBOLS, MAXLAYER, MAXCACHEFRAME,  # noqa: F401
                     Rendezvous, device_count, device_pci_bus_id)
from .weights_init import glorot_net  # noqa: F401
