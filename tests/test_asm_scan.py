"""Guard against loads that the compiler serialises (load, s_waitcnt vmcnt(0), use -- once per load): round 6 found three launches
like that in the listing (the bf16 dgrad's prologue, the output layer's slab reduce, the bf16-segment exchange kernel) and one
experiment whose main loop was drained in front of every LDS read (DESIGN.md, item 9 of the round-6 list).  This test compiles the
device code of the library to assembly (hipcc cross-compiles gfx950 without a GPU) and bounds the number of full drains in the kernels
where a drain per load would cost the most.  The bounds are the counts of the shipped kernels plus slack; tools/asm_waitcnt_scan.sh
prints the whole table."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

# mangled-name fragment -> most full drains (s_waitcnt vmcnt(0)) the kernel may contain
BOUNDS = {
    "bp_step": {
        "_Z12bp_wgrad_dmaILi16ELi4ELi4ELi256ELb0EE": 6,          # fused wgrad + update (3: epilogue + the explicit drains)
        "_Z12bp_wgrad_dmaILi16ELi4ELi4ELi256ELb1EE": 6,          # data-parallel store form
        "_Z12bp_gemm_bf16ILi2ELi128ELb0ELb1ELi1EE": 6,           # bf16 dgrad, LDS-DMA form (was 34: 32 serial prologue loads)
        "_Z12bp_gemm_bf16ILi2ELi128ELb0ELb0ELi1EE": 6,
        "_Z12bp_gemm_bf16ILi2ELi64ELb0ELb0ELi1EE": 5,
        "_Z12bp_gemm_bf16ILi2ELi32ELb0ELb0ELi1EE": 5,
        "_Z12bp_gemm_bf16ILi0ELi128ELb1ELb1ELi1EE": 5,           # bf16 forward, LDS-DMA form
        "_Z21bp_wgrad_dma_bf16_sixILi512ELi64ELi3EE": 7,
        "_Z13bp_gemm_multiI10GemmKernelILi32ELi64ELi128ELi1ELi2ELb1ELb1ELi2EEE": 5,   # hidden dgrad
    },
    "bp_dp": {
        "_Z19bp_dp_reduce_updateILi0ELb1EE": 8,                  # bf16 gradient segments (was 36 for 41 loads)
        "_Z19bp_dp_reduce_updateILi0ELb0EE": 8,
        "_Z19bp_dp_reduce_updateILi8ELb0EE": 8,
    },
}


def _drains(unit, tmp):
    out = os.path.join(tmp, unit + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                           os.path.join(CSRC, unit + ".hip")], stderr=subprocess.DEVNULL)
    counts, cur = {}, None
    for line in open(out):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            cur = m.group(1); counts[cur] = 0
        elif cur and "s_waitcnt vmcnt(0)" in line:
            counts[cur] += 1
    return counts


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("unit", sorted(BOUNDS))
def test_hot_kernels_have_no_load_by_load_drains(unit, tmp_path):
    counts = _drains(unit, str(tmp_path))
    for frag, bound in BOUNDS[unit].items():
        hits = {k: v for k, v in counts.items() if k.startswith(frag)}
        assert hits, "kernel %s not found in %s (renamed? update the table)" % (frag, unit)
        for k, v in hits.items():
            assert v <= bound, "%s: %d full vmcnt drains (bound %d): loads serialised by the compiler?" % (k, v, bound)
    shutil.rmtree(str(tmp_path), ignore_errors=True)
