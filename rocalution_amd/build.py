"""Build librocalution_amd.so (hipcc, gfx950 only) in-tree.

    python -m rocalution_amd.build [--force]

Every .hip / .cpp under rocalution_amd/csrc is compiled to an object under csrc/_obj and linked into
rocalution_amd/librocalution_amd.so.  hipcc cross-compiles without a GPU, so this is also the
"does it build" check of __graft_entry__.build().
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# RAMD_BUILD_FLAVOUR=asan: a second, separate library (librocalution_amd_asan.so, objects under csrc/_obj_asan) whose HOST code
# -- the ABI layer and the C++ API layer of include/rocalution behind it -- is instrumented by AddressSanitizer; device code is
# left alone (-fno-gpu-sanitize).  The counterpart of the reference's BUILD_ADDRESS_SANITIZER option (CMakeLists.txt:83-90).
# Use:  RAMD_BUILD_FLAVOUR=asan python -m rocalution_amd.build ;  RAMD_LIB=.../librocalution_amd_asan.so
#       LD_PRELOAD=$(hipcc --print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -m gpu
# State: builds and loads (python under the preloaded runtime imports it); on this image the first device allocation then
# fails inside the sanitizer's hsa_amd_memory_pool_allocate interceptor ("out of memory", gpurun_out/r03bn) -- the ROCm
# installation here has no ASAN-instrumented runtime libraries (no /opt/rocm/lib/asan), which that interceptor expects.
FLAVOUR = os.environ.get("RAMD_BUILD_FLAVOUR", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + FLAVOUR if FLAVOUR else ""))
LIB = os.path.join(HERE, "librocalution_amd" + ("_" + FLAVOUR if FLAVOUR else "") + ".so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

# -ffp-contract=off: element-wise results must equal the reference's host expressions bit for bit
# (x86-64 baseline builds of the reference do not fuse multiply-add); the kernels are bandwidth-bound
# so FMA contraction would buy nothing.
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
            "-I" + os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("RAMD_EXTRA_CXXFLAGS", "").split()
LDFLAGS = ["-shared", "-fPIC", "--offload-arch=gfx950", "-L" + os.path.join(ROCM, "lib"), "-lrccl",
           "-Wl,-rpath," + os.path.join(ROCM, "lib")]
if FLAVOUR == "asan":
    _SAN = ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-g", "-fno-omit-frame-pointer"]
    CXXFLAGS = [f for f in CXXFLAGS if f != "-O3"] + ["-O1"] + _SAN
    LDFLAGS = LDFLAGS + _SAN
elif FLAVOUR:
    raise SystemExit("RAMD_BUILD_FLAVOUR: only 'asan' is known")


def _sources():
    srcs = []
    for root, _, files in os.walk(CSRC):
        if os.path.basename(root) == "_obj":
            continue
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                srcs.append(os.path.join(root, f))
    return srcs


def _headers_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for r, _, files in os.walk(root):
            for f in files:
                if f.endswith((".hpp", ".h")):
                    m = max(m, os.path.getmtime(os.path.join(r, f)))
    return m


def _compile(src, obj):
    cmd = [HIPCC] + CXXFLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, p.returncode, p.stdout


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hm = _headers_mtime()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            for src, rc, out in ex.map(lambda j: _compile(*j), jobs):
                if verbose or rc != 0:
                    sys.stderr.write(out)
                if rc != 0:
                    raise RuntimeError("hipcc failed on " + src)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC] + objs + LDFLAGS + ["-o", LIB]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stdout)
            raise RuntimeError("link of librocalution_amd.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
