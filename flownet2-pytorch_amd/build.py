"""Builds libflownet2_hip.so (hand-written gfx950 HIP kernels + C ABI) and the three pybind
modules correlation_cuda / resample2d_cuda / channelnorm_cuda, in-tree, without hipify.

  python flownet2-pytorch_amd/build.py            # everything
  python flownet2-pytorch_amd/build.py --lib      # kernels + C ABI only (seconds)

hipcc cross-compiles for gfx950 without a GPU.  The pybind modules are plain C++ (g++) against
the ATen headers; they never see a kernel and are not run through torch's hipify pass.
"""
import argparse
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libflownet2_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNEL_SRCS = ["capi.hip", "channelnorm.hip", "resample2d.hip", "correlation_direct.hip", "correlation_mfma.hip", "correlation_f16x2.hip", "correlation_f16x2_wide.hip", "correlation_f16_fwd.hip", "correlation_f16_bwd.hip", "correlation_f16x2_bwd.hip", "correlation_fused_bwd.hip", "correlation_f16x2_bwd_wide.hip", "correlation_mfma_bwd.hip", "correlation_mfma_f64.hip",
               "multiscale_loss.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-munsafe-fp-atomics", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
MODULES = ["correlation_cuda", "resample2d_cuda", "channelnorm_cuda", "multiscale_loss_cuda"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build step failed: " + cmd[0])
    return r


def build_lib(force=False, debug=False):
    """libflownet2_hip.so (the product: public C ABI only) or, with debug=True, libflownet2_hip_debug.so: the same kernels plus
    the fn2_debug_* entry points and the profiling instantiations they select (csrc/fn2_debug.h; scripts/ and ablation runs)."""
    objdir = os.path.join(LIBDIR, "debug") if debug else LIBDIR
    lib = os.path.join(LIBDIR, "libflownet2_hip_debug.so") if debug else LIB
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "flownet2_hip.h"))
    objs, jobs = [], []
    for src in KERNEL_SRCS:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or not _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + HIP_FLAGS + (["-DFN2_DEBUG_BUILD"] if debug else []) + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:   # the translation units are independent: compile them side by side (each hipcc is single-threaded)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(_run, jobs))
    if force or not _newer(lib, objs + [os.path.join(CSRC, "exports.map")]):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", lib] + objs)
    return lib


def build_modules(force=False):
    import torch
    tdir = os.path.dirname(torch.__file__)
    tinc = [os.path.join(tdir, "include"), os.path.join(tdir, "include", "torch", "csrc", "api", "include")]
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    outs, jobs = [], []
    for m in MODULES:
        src = os.path.join(CSRC, "binding", m + ".cpp")
        out = os.path.join(HERE, m + ext)
        deps = [src, os.path.join(CSRC, "binding", "binding_common.h"), os.path.join(HERE, "..", "include", "flownet2_hip.h")]
        if force or not _newer(out, deps):
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                   "-DTORCH_EXTENSION_NAME=" + m, "-DTORCH_API_INCLUDE_EXTENSION_H",
                   "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-I" + os.path.join(HERE, "..", "include"),
                   "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
            cmd += ["-I" + i for i in tinc]
            cmd += [src, "-o", out, "-L" + os.path.join(tdir, "lib"), "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
                    "-ltorch_hip", "-ltorch_python", "-L" + LIBDIR, "-lflownet2_hip",
                    "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath," + os.path.join(tdir, "lib")]
            jobs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        outs.append(out)
    for cmd, pr in jobs:  # the three translation units compile concurrently
        log, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + log)
            raise RuntimeError("build step failed: g++")
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="store_true", help="only the kernel library")
    ap.add_argument("--no-debug", action="store_true", help="skip libflownet2_hip_debug.so (profiling entry points)")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    if a.no_debug:
        print(build_lib(a.force))
    else:   # the two libraries share no object files: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as pool:
            futs = [pool.submit(build_lib, a.force, False), pool.submit(build_lib, a.force, True)]
            for f in futs:
                print(f.result())
    if not a.lib:
        for o in build_modules(a.force):
            print(o)


if __name__ == "__main__":
    main()
