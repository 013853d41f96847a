"""Builds the native parts of mad_icp_amd in-tree (the .so files travel with the repo snapshot).

  libmadicp_hip.so   hipcc, gfx950 only: HIP kernels + the C ABI of include/madicp_hip.h
  libmadicp_host.so  g++: host MAD-tree builder, C ABI of include/madicp_host.h
  pybind/*.so        g++ + pybind11: pyvector, pymadtree, pymadicp, pypeline (the reference's module names)

No CPU fallback is ever built for the HIP entry points.
"""
import hashlib
import os
import subprocess
import sys
import sysconfig

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(PKG, "csrc")

# MADICP_NATIVE_DIR: build into (and load from, see capi._load) another directory — used by tests/test_redux_variant.py to
# build the whole stack a second time with other defines without touching the in-tree libraries
OUT = os.environ.get("MADICP_NATIVE_DIR") or PKG

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wno-unused-value"]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra", "-fopenmp"]
# MADICP_EXTRA_DEFINES="-DMADICP_REDUX_SCALAR_ONLY": the Eigen-3.3 evaluation order of 3-vector reductions, in the
# product AND (oracle/Makefile reads the same variable) in the oracle — tests/test_redux_variant.py builds both that way
_EXTRA = os.environ.get("MADICP_EXTRA_DEFINES", "").split()
EXTRA_HIP_FLAGS = list(_EXTRA)
EXTRA_HOST_FLAGS = list(_EXTRA)


def source_hash(sources, flags=()):
    """sha-256 over the build command's flags and the CONTENT of every source / header it depends on."""
    h = hashlib.sha256()
    for f in flags:
        h.update(str(f).encode() + b"\0")
    for s in sorted(sources):
        h.update(os.path.relpath(s, ROOT).encode() + b"\0")
        with open(s, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale(target, sources, flags=()):
    """A target is up to date iff it exists and was built from exactly these sources with these flags (content hash
    kept beside it in <target>.srchash) — modification times say nothing once a snapshot has been copied to another box."""
    if not os.path.exists(target):
        return True
    try:
        with open(target + ".srchash") as fh:
            return fh.read().strip() != source_hash(sources, flags)
    except OSError:
        return True


def _stamp(target, sources, flags=()):
    with open(target + ".srchash", "w") as fh:
        fh.write(source_hash(sources, flags) + "\n")


def _run(cmd):
    print("[mad_icp_amd build]", " ".join(cmd), file=sys.stderr, flush=True)  # (stdout belongs to bench.py's JSON line)
    subprocess.check_call(cmd, stdout=sys.stderr)


def _glob(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def _hip_deps(srcs):
    return (srcs + _glob(os.path.join(CSRC, "hip"), (".h",)) + _glob(os.path.join(CSRC, "common"), (".h",)) +
            _glob(INC, (".h",)))


def build_hip(force=False):
    out = os.path.join(OUT, "libmadicp_hip.so")
    srcs = [os.path.join(CSRC, "hip", "madicp_capi.hip")]
    deps = _hip_deps(srcs)
    flags = HIP_FLAGS + EXTRA_HIP_FLAGS
    if force or _stale(out, deps, flags):
        _run([HIPCC] + flags + ["-I" + INC, "-I" + os.path.join(CSRC, "hip")] + srcs + ["-o", out, "-lrccl"])
        _stamp(out, deps, flags)
    return out


def hip_source_hash():
    srcs = [os.path.join(CSRC, "hip", "madicp_capi.hip")]
    return source_hash(_hip_deps(srcs), HIP_FLAGS + EXTRA_HIP_FLAGS)


HOST_SRCS = ("tree_builder.cpp", "host_capi.cpp", "mad_tree.cpp", "mad_icp.cpp", "vel_estimator.cpp", "pipeline.cpp", "deskew.cpp")


def build_host(force=False):
    """libmadicp_host.so: tree builder + the host classes (MADtree, MADicp, VelEstimator, Pipeline).  Links
    against libmadicp_hip.so — the host classes have no other implementation to call."""
    out = os.path.join(OUT, "libmadicp_host.so")
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in HOST_SRCS]
    deps = srcs + _glob(hdir, (".h",)) + _glob(os.path.join(CSRC, "common"), (".h",)) + _glob(INC, (".h",))
    flags = HOST_FLAGS + EXTRA_HOST_FLAGS
    if force or _stale(out, deps, flags):
        _run([CXX] + flags + ["-shared", "-I" + INC, "-I" + hdir] + srcs +
             ["-o", out, "-L" + OUT, "-lmadicp_hip", "-Wl,-rpath,$ORIGIN", "-pthread"])
        _stamp(out, deps, flags)
    return out


def build_pybind(force=False):
    import pybind11

    hdir = os.path.join(CSRC, "host")
    pdir = os.path.join(CSRC, "pybind")
    outdir = os.path.join(OUT, "pybind")
    os.makedirs(outdir, exist_ok=True)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    deps_h = (_glob(hdir, (".h",)) + _glob(pdir, (".h",)) + _glob(os.path.join(CSRC, "common"), (".h",)) +
              _glob(INC, (".h",)))
    flags = HOST_FLAGS + EXTRA_HOST_FLAGS
    built = []
    for mod in ("pyvector", "pymadtree", "pymadicp", "pypeline"):
        src = os.path.join(pdir, mod + ".cpp")
        out = os.path.join(outdir, mod + suffix)
        if force or _stale(out, [src] + deps_h, flags):
            _run([CXX] + flags + ["-shared", "-fvisibility=hidden", "-I" + INC, "-I" + hdir, "-I" + pdir,
                                  "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], src,
                                  "-o", out, "-L" + OUT, "-lmadicp_host", "-lmadicp_hip", "-Wl,-rpath,$ORIGIN/..",
                                  "-pthread"])
            _stamp(out, [src] + deps_h, flags)
        built.append(out)
    return built


def build_all(force=False):
    return [build_hip(force), build_host(force)] + build_pybind(force)


# The MEASUREMENT build: the same sources with -DMADICP_MEASURE — include/madicp_hip_measure.h's timing / calibration / test
# aids and the realtime rule's test seam in pypeline — into mad_icp_amd/_measure/ (in-tree, git-ignored like every built file,
# travels to the GPU box).  The default build above is the product: it exports include/madicp_hip.h and nothing else.
# bench.py's roofline legs and the few tests that look inside a device tree build load it through capi.measure_variant().
MEASURE_DIR = os.path.join(PKG, "_measure")


def build_measure(force=False):
    if os.path.abspath(OUT) == os.path.abspath(MEASURE_DIR):
        return build_all(force)
    os.makedirs(os.path.join(MEASURE_DIR, "pybind"), exist_ok=True)
    env = dict(os.environ, MADICP_NATIVE_DIR=MEASURE_DIR,
               MADICP_EXTRA_DEFINES=" ".join(_EXTRA + ["-DMADICP_MEASURE"]), PYTHONPATH=os.pathsep.join([ROOT] + sys.path))
    # (force: the HIP library only — it holds the kernels a measurement is about; host library and bindings by content hash)
    subprocess.check_call([sys.executable, "-c", "from mad_icp_amd import _build; _build.build_hip(%r); _build.build_host(); "
                           "_build.build_pybind()" % bool(force)], env=env, cwd=ROOT, stdout=sys.stderr)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return [os.path.join(MEASURE_DIR, "libmadicp_hip.so"), os.path.join(MEASURE_DIR, "libmadicp_host.so")] + [
        os.path.join(MEASURE_DIR, "pybind", m + suffix) for m in ("pyvector", "pymadtree", "pymadicp", "pypeline")]


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    if "--no-measure" not in sys.argv and "-DMADICP_MEASURE" not in _EXTRA:
        build_measure(force="--force" in sys.argv)
