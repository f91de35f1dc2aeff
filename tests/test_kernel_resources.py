"""The search kernel is compiled for a fixed register budget (three waves per SIMD at d <= 61); one extra live value
spills to scratch and halves its speed (measured: 64 -> 129 ms at 1M cells).  Compile the product instantiations and
check what hipcc reports: no scratch, the intended occupancy.  (CPU only: hipcc cross-compiles without a GPU.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resource_usage(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", os.devnull]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200).stdout
    rows, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:"):
            if cur and key in line:
                rows[cur][key] = int(line.split(key)[1].split()[0])
    return rows


@pytest.mark.timeout(1500)
def test_search_kernels_do_not_spill():
    rows = _resource_usage(os.path.join(ROOT, "meld_amd", "csrc", "knn16.hip"))
    product = {k: v for k, v in rows.items() if "knn16_topk_kernelILi" in k and "ELi0ELi" in k}  # ABL = 0
    # KB = 1..9 x NPROD in {1, 3}, table-driven (LIST = false) + KB = 1..9 list-driven hi-only first pass (LIST = true)
    assert len(product) == 27 and sum("ELb1E" in k for k in product) == 9, sorted(product)
    for name, r in product.items():
        assert r["ScratchSize [bytes/lane]:"] == 0, (name, r)
    # the benchmark configuration (d = 50: KB = 4, hi-only first pass) keeps three waves per SIMD, on both kernels
    for first_pass in [v for k, v in product.items() if "ILi4ELi0ELi1E" in k]:
        assert first_pass["Occupancy [waves/SIMD]:"] == 3 and first_pass["VGPRs:"] <= 168, first_pass
    bounds = [v for k, v in rows.items() if "knn16_tile_bounds_kernel" in k]
    assert bounds and all(v["ScratchSize [bytes/lane]:"] == 0 for v in bounds)


def test_recurrence_kernel_keeps_its_stream_slots_to_itself():
    """The consumer stream of pt_step_kernel (csrc/spmm_tiled.hip) keeps loads in flight in physical registers
    v96..v119 that only its inline asm may name: between the first and the last hand-counted wait of the consumer loop
    no compiler-generated instruction may touch them (the allocator is free to use them elsewhere, e.g. in the loader
    waves' branch, which runs no stream).  Checked on the assembly hipcc emits for every instantiation."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "meld_amd", "csrc", "spmm_tiled.hip")
    asm = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "--cuda-device-only",
                          "-S", src, "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=1200).stdout
    slot = re.compile(r"\bv\[?(9[6-9]|1[01][0-9])\b")
    kernels = 0
    fn, lines = None, []
    for line in asm.splitlines() + ["_end:"]:
        m = re.match(r"^(_Z\w*pt_step_kernel\w*):", line)
        if m or line.startswith("_end:") or (fn and ".Lfunc_end" in line):
            if fn:
                # basic blocks of the consumer loop: the block that holds the first hand-counted wait names the loop
                # header in its label comment; every block whose label refers to that header is part of the loop
                starts = [i for i, l in enumerate(lines) if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)]
                in_asm, waits = False, []  # the hand-counted waits are the ones inside inline-asm blocks
                for i, l in enumerate(lines):
                    if "#ASMSTART" in l:
                        in_asm = True
                    elif "#ASMEND" in l:
                        in_asm = False
                    elif in_asm and "s_waitcnt vmcnt(14)" in l:
                        waits.append(i)
                wait_set = set(waits)
                assert len(waits) >= 8, (fn, len(waits))
                votes = {}
                for wi in waits:  # (a peeled copy of an iteration may sit outside the loop: take the loop most waits are in)
                    label = lines[max(i for i in starts if i <= wi)]
                    m2 = re.search(r"Header=(BB\d+_\d+)", label) or (re.match(r"^\.L(BB\d+_\d+):", label) if "Loop Header" in label else None)
                    if m2:
                        votes[m2.group(1)] = votes.get(m2.group(1), 0) + 1
                header = max(votes, key=votes.get)
                assert votes[header] >= 7, votes
                checked = 0
                for bi, i0 in enumerate(starts):
                    i1 = starts[bi + 1] if bi + 1 < len(starts) else len(lines)
                    label = lines[i0]
                    in_loop = re.search(r"(Header=|Loop |^\.L)" + header + r"\b", label) is not None
                    if not in_loop and not any(i in wait_set for i in range(i0, i1)):
                        continue
                    inside = False
                    for l in lines[i0 + 1 : i1]:
                        if "#ASMSTART" in l:
                            inside = True
                        elif "#ASMEND" in l:
                            inside = False
                        elif not inside and not l.lstrip().startswith(";"):
                            assert not slot.search(l.split(";")[0]), (fn, l)
                            checked += 1
                assert checked > 200, (fn, checked)
                kernels += 1
            fn, lines = (m.group(1) if m else None), []
        elif fn:
            lines.append(line)
    assert kernels == 3, kernels  # <2, fp64>, <1, fp64>, <1, fp32 values>
