"""k_entropy decodes one packet per wave with WAVE-UNIFORM control flow: its bit reader, tables and loop counters belong in
scalar registers.  Whether they are there is the compiler's decision, and it is easily lost -- one lane-masked store in
lw_dev_entropy.h (an `if (lane < n)`) made the compiler treat the values merged behind that branch as divergent, and the
whole reader with every loop around it moved to vector registers and exec masks (3-4 x the instructions, same results: no
parity test notices).  This test compiles the kernel for gfx950 and counts the exec-mask manipulations in its code: a
scalar build has ~20 (the zeroing / write-out loops, the per-lane selects of the floor curve), a vectorised one hundreds."""
import os
import re
import shutil
import subprocess

import pytest

from common import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_entropy_kernel_control_flow_is_scalar(tmp_path):
    src = os.path.join(ROOT, "lewton_amd", "csrc", "lw_kernels_entropy.hip")
    out = str(tmp_path / "ent.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "lewton_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    kernels = re.split(r"\n(?=_Z9k_entropyILb[01]E)", text)[1:]
    assert len(kernels) == 2
    for k in kernels:
        body = k.split("s_endpgm")[0]
        masks = len(re.findall(r"\bs_(?:and|andn2|or|xor)_saveexec_b64\b", body))
        assert masks <= 40, "exec-mask manipulations in %s: %d -- the packet's control flow has moved to the vector unit" % (
            k[:24], masks)
        # the hand-written loops are in the build (device path), scratch memory is not used
        assert "ds_read_u16" in body and "s_load_dwordx8" in body
    assert re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text) == ["0", "0"]
