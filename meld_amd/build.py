"""Build libmeld_hip.so in-tree for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

from ._options import is_set, opt

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["api.hip", "knn.hip", "knn16.hip", "refine.hip", "assemble.hip", "spmm.hip", "spmm_tiled.hip", "reorder.hip", "kmeans.hip", "labels.hip", "sharded.hip", "frame.hip"]
OUT = os.path.join(_HERE, "libmeld_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-file additions.  knn16.hip: the minima over MFMA accumulators (tile bounds, seeds) are fmin chains on values the compiler
# cannot prove canonical, and with NaNs honoured it quiets every one of them first (v_max x, x: 25 of the 266 vector instructions
# of the bounds kernel's loop, which is bound by those).  Nothing in that file produces or tests for a NaN -- inputs are checked
# finite on the host, paddings are +inf by construction, never inf - inf -- and the graph is bit-identical with and without.
FILE_FLAGS = {"knn16.hip": ["-fno-honor-nans"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required to build meld_amd)")
    return exe


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def check_stream_slots(asm):
    """Assembly guard of csrc/spmm_tiled.hip, a GATE of the build: the consumer stream of pt_step_kernel keeps loads in flight in
    physical registers v96..v119 that only its inline asm may name (the kernel is capped at v0..v95, hipcc warns that the asm
    clobbers reserved registers -- that is the point).  Between the first and the last hand-counted wait of the consumer loop no
    compiler-generated instruction may touch them; a toolchain that allocates them there would make the kernel return stale
    data.  ``asm``: the text of ``hipcc -S`` for the file.  Returns the number of kernels checked; raises AssertionError."""
    import re

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
    return kernels


def gate_stream_slots(hipcc=None, verbose=True):
    """Compile csrc/spmm_tiled.hip to assembly with the build's flags and run ``check_stream_slots`` on it; RuntimeError if the
    guard fails (the library is then NOT linked)."""
    hipcc = hipcc or _hipcc()
    src = os.path.join(CSRC, "spmm_tiled.hip")
    cmd = [hipcc] + FLAGS + ["--cuda-device-only", "-S", src, "-o", "-"]
    if verbose:
        print("[meld_amd.build] gate:", " ".join(cmd), flush=True)
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if run.returncode != 0 or not run.stdout:
        raise RuntimeError("meld_amd.build: `hipcc -S` of spmm_tiled.hip failed (status {}), the register-slot guard of pt_step_kernel "
                           "could not run -- not linking:\n{}".format(run.returncode, run.stderr[-2000:]))
    asm = run.stdout
    try:
        return check_stream_slots(asm)
    except (AssertionError, ValueError, StopIteration) as e:  # (ValueError: no loop-header label matched -- another label format)
        raise RuntimeError("meld_amd.build: the register-slot guard of pt_step_kernel failed ({}); the tiled recurrence kernel "
                           "cannot be trusted with this toolchain -- not linking".format(e))


def build(force=False, verbose=True):
    """Compile every HIP source to an object (in parallel) and link the shared library."""
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(_HERE, "..", "include", "meld_hip.h")]
    procs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print("[meld_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    rebuilt = [src for src, _ in procs]
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on {}".format(src))
    if "spmm_tiled.hip" in rebuilt or (force and opt("MELD_BUILD_SKIP_GATE") != "1"):
        # the recurrence kernel names physical registers its inline asm owns: checked on the emitted assembly before linking
        if opt("MELD_BUILD_SKIP_GATE") != "1":
            gate_stream_slots(hipcc, verbose)
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT]
        if verbose:
            print("[meld_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
