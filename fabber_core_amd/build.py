"""In-tree build of the native libraries (hipcc for gfx950, g++ for the host side).

    python -m fabber_core_amd.build [--force] [--jobs N]

Produces fabber_core_amd/lib/libfabber_vb_hip.so (HIP kernels + fabber_vb C ABI). The objects
are compiled one translation unit per process so that the template instantiations build in
parallel; nothing is cached outside the repository, so the built .so travels with a snapshot
of the tree.
"""
import argparse
import concurrent.futures
import hashlib
import re
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
BINDIR = os.path.join(HERE, "bin")
OBJDIR = os.path.join(HERE, "build", "obj")
ARCH = "gfx950"

HIP_SOURCES = ["vb_api.hip", "vb_lane_poly.hip", "vb_lane_linear.hip", "vb_lane_exp.hip", "vb_lane_wide.hip", "vb_lane_pattern_poly_2.hip", "vb_lane_pattern_poly_4.hip",
               "vb_lane_pattern_linear_2.hip", "vb_lane_pattern_linear_4.hip", "vb_lane_pattern_exp_2.hip", "vb_lane_pattern_exp_4.hip", "vb_lane_ar_poly.hip",
               "vb_lane_ar_linear.hip", "vb_lane_ar_exp.hip", "vb_lane_arn_linear.hip", "vb_lane_arn_more.hip", "vb_spatial_api.hip", "vb_spatial_poly.hip",
               "vb_spatial_linear.hip", "vb_spatial_exp.hip", "vb_spatial_host.hip", "vb_spatial_more.hip",
               "vb_spatial_nz_poly.hip", "vb_spatial_nz_linear.hip", "vb_spatial_nz_linear2.hip", "vb_spatial_nz_exp.hip", "vb_spatial_nz_host.hip",
               "vb_spatial_nz_arn.hip", "vb_spatial_nz_arn_linear2.hip", "vb_spatial_nz_arn_linear3.hip", "vb_spatial_nz_arn_linear4.hip", "vb_spatial_nz_arn_poly.hip",
               "vb_spatial_nz_arn_exp2.hip", "vb_spatial_nz_arn_exp4.hip", "vb_spatial_nz_p8_linear.hip", "vb_spatial_nz_p8_poly.hip", "vb_spatial_nz_p8_exp.hip", "vb_wave.hip", "vb_hostmodel_api.hip", "vb_nlls.hip"]
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-fast-math",
             "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the fabber_vb HIP engine cannot be built")
    return exe


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "fabber_vb.h"))
    return hs


_INCLUDE = re.compile(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', re.M)


def _closure(src, _seen=None):
    """src and the project headers it includes, transitively (every project header is included with quotes, relative
    to the including file, csrc/ or include/): a change to one kernel header rebuilds the sources that see it, not all
    fifty (a full build is ~25 minutes on 8 cores)."""
    seen = _seen if _seen is not None else {}
    if src in seen:
        return seen
    seen[src] = True
    with open(src, "r", errors="replace") as fh:
        text = fh.read()
    for inc in _INCLUDE.findall(text):
        for d in (os.path.dirname(src), CSRC, os.path.join(os.path.dirname(HERE), "include")):
            cand = os.path.normpath(os.path.join(d, inc))
            if os.path.exists(cand):
                _closure(cand, seen)
                break
    return seen


def _stamp(src, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for f in sorted(_closure(src)):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def restamp():
    """Write the stamps of the objects that exist as if they had just been built (after changing how stamps are formed)."""
    flags = HIP_FLAGS + os.environ.get("FVB_EXTRA_HIPCC_FLAGS", "").split()
    for s in HIP_SOURCES:
        obj = os.path.join(OBJDIR, s + ".o")
        if os.path.exists(obj):
            with open(obj + ".stamp", "w") as fh:
                fh.write(_stamp(os.path.join(CSRC, s), flags))


def _compile_one(args):
    src, flags, force = args
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False, ""
    cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True, p.stderr


def build_hip(force=False, jobs=None, verbose=True, extra_flags=(), variant=None, only=None):
    """variant: build an experiment copy of the engine - objects under build/obj_<variant>, library
    lib/libfabber_vb_hip_<variant>.so (load it with FVB_LIB_PATH=...) - next to the product library.
    only: with a variant, the sources the extra flags apply to; the other objects are the product's."""
    global OBJDIR
    objdir_product = OBJDIR
    if variant:
        OBJDIR = os.path.join(HERE, "build", "obj_" + variant)
    try:
        return _build_hip(force, jobs, verbose, extra_flags, variant, only, objdir_product)
    finally:
        OBJDIR = objdir_product


def _build_hip(force, jobs, verbose, extra_flags, variant, only=None, objdir_product=None):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    flags = HIP_FLAGS + list(extra_flags) + os.environ.get("FVB_EXTRA_HIPCC_FLAGS", "").split()
    sources = [s for s in HIP_SOURCES if not (variant and only) or s in only]
    work = [(os.path.join(CSRC, s), flags, force) for s in sources]
    jobs = jobs or min(len(work), os.cpu_count() or 4)
    objs, rebuilt = [], False
    if variant and only:
        for s in HIP_SOURCES:
            if s not in only:
                obj = os.path.join(objdir_product, s + ".o")
                if not os.path.exists(obj):
                    raise RuntimeError("build the product library first: %s is missing" % obj)
                objs.append(obj)
        rebuilt = True
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did, warn in ex.map(_compile_one, work):
            objs.append(obj)
            rebuilt |= did
            if verbose and did:
                print("[build] compiled", os.path.basename(obj), file=sys.stderr)
            if verbose and warn.strip():
                print(warn, file=sys.stderr)
    lib = os.path.join(LIBDIR, "libfabber_vb_hip%s.so" % ("_" + variant if variant else ""))
    if rebuilt or not os.path.exists(lib):
        cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
        if verbose:
            print("[build] linked", lib, file=sys.stderr)
    return lib


HOST_DIR = os.path.join(CSRC, "host")
HOST_FLAGS = ["-O2", "-g", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-parameter", "-Wno-deprecated-declarations"]


def _host_sources():
    return sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cc"))


def _compile_host_one(args):
    src, force = args
    obj = os.path.join(OBJDIR, "host_" + os.path.basename(src) + ".o")
    stamp_file = obj + ".stamp"
    h = hashlib.sha256()
    h.update(" ".join(HOST_FLAGS).encode())
    deps = [src] + _headers() + [os.path.join(d, f) for d in (os.path.join(HOST_DIR, "fabber_core"), os.path.join(HOST_DIR, "armawrap"))
                                 for f in sorted(os.listdir(d))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "fabber_capi.h"))
    for f in deps:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = h.hexdigest()
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = ["g++"] + HOST_FLAGS + ["-I", HOST_DIR, "-I", os.path.join(HOST_DIR, "fabber_core"), "-c", src, "-o", obj]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build_host(force=False, jobs=None, verbose=True):
    """libfabbercore_amd.so: the reference's plugin classes + fabber_capi, linked against the
    HIP engine library that sits next to it (rpath $ORIGIN)."""
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    work = [(s, force) for s in _host_sources()]
    objs, rebuilt = [], False
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
        for obj, did in ex.map(_compile_host_one, work):
            objs.append(obj)
            rebuilt |= did
            if verbose and did:
                print("[build] compiled", os.path.basename(obj), file=sys.stderr)
    lib = os.path.join(LIBDIR, "libfabbercore_amd.so")
    engine = os.path.join(LIBDIR, "libfabber_vb_hip.so")
    if rebuilt or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(engine):
        cmd = ["g++", "-shared", "-fPIC", "-o", lib] + objs + ["-L", LIBDIR, "-lfabber_vb_hip", "-ldl", "-lz", "-Wl,-rpath,$ORIGIN",
                                                                "-Wl,--no-undefined"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
        if verbose:
            print("[build] linked", lib, file=sys.stderr)
    return lib


def build_cli(force=False, verbose=True):
    """bin/fabber: the command line tool (fabber_main.cc), linked against libfabbercore_amd.so."""
    os.makedirs(BINDIR, exist_ok=True)
    exe = os.path.join(BINDIR, "fabber")
    src = os.path.join(HOST_DIR, "cli", "fabber_main.cc")
    lib = os.path.join(LIBDIR, "libfabbercore_amd.so")
    if force or not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(lib), os.path.getmtime(src)):
        cmd = ["g++"] + HOST_FLAGS + ["-I", HOST_DIR, src, "-o", exe, "-L", LIBDIR, "-lfabbercore_amd", "-lfabber_vb_hip",
                                      "-Wl,-rpath,$ORIGIN/../lib"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
        if verbose:
            print("[build] linked", exe, file=sys.stderr)
    return exe


def build_all(force=False, jobs=None, verbose=True):
    libs = [build_hip(force=force, jobs=jobs, verbose=verbose)]
    libs.append(build_host(force=force, jobs=jobs, verbose=verbose))
    libs.append(build_cli(force=force, verbose=verbose))
    return libs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--variant", default=None, help="experiment copy of the HIP engine only (with --flags)")
    ap.add_argument("--flags", default="", help="extra hipcc flags for --variant, e.g. --flags=-DFVB_NO_RESCUE")
    ap.add_argument("--only", default="", help="with --variant: comma-separated sources the flags apply to (the rest: the product's objects)")
    a = ap.parse_args()
    if a.variant:
        print(build_hip(force=a.force, jobs=a.jobs, extra_flags=a.flags.split(), variant=a.variant,
                        only=[x for x in a.only.split(",") if x] or None))
        sys.exit(0)
    for l in build_all(force=a.force, jobs=a.jobs):
        print(l)
