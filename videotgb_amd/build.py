"""Build libvtgb.so (HIP, gfx950 only) in-tree with hipcc.  `python -m videotgb_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvtgb.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "gemm_h8.hip", "conv64.hip", "conv_f32.hip", "attn.hip", "elementwise.hip", "select.hip", "forward.hip", "llm.hip", "raft.hip", "raft_x3.hip", "gru_fused.hip", "raft_corr.hip", "raft_enc.hip", "train.hip", "train_attn.hip", "train_ops.hip", "comm.hip"]


FLAGS_STAMP = os.path.join(HERE, "build", "flags")


def _flags():
    return ("debug-hooks" if os.environ.get("VTGB_DEBUG_HOOKS") == "1" else "production") + os.environ.get("VTGB_CFLAGS", "")


def _stale():
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(FLAGS_STAMP) or open(FLAGS_STAMP).read() != _flags():   # a debug-hook build is never mistaken for the product
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "vtgb.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into videotgb_amd/libvtgb.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
               "-I", os.path.join(REPO, "include"), "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        cmd[1:1] = os.environ.get("VTGB_CFLAGS", "").split()      # experiments: e.g. VTGB_CFLAGS=-DVTGB_SPREAD=0
        if os.environ.get("VTGB_DEBUG_HOOKS") == "1":      # experiment knobs + timing-only ablation kernels (tools/gemm_ablate.py)
            cmd.insert(1, "-DVTGB_DEBUG_HOOKS")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            sys.stderr.write(f"--- {src}\n{out}\n")
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed building libvtgb.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flags())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
