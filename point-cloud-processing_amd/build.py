"""Builds libpcpx.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container too.  The .so is kept
next to this file: it is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libpcpx.so")
SOURCES = ["pcpx_query.hip", "pcpx_few.hip", "pcpx_range.hip", "pcpx_filter.hip", "pcpx_normals.hip", "pcpx_prep.hip", "pcpx_orient.hip", "pcpx_build.hip", "pcpx_shard.hip", "pcpx_sort.hip",
           "pcpx_comm.hip", "pcpx_kd.hip", "pcpx_api.hip"]
HEADERS = [os.path.join(CSRC, "pcpx_internal.h"), os.path.join(CSRC, "pcpx_device.h"), os.path.join(CSRC, "pcpx_eig3.h"), os.path.join(CSRC, "pcpx_curve.h"),
           os.path.join(CSRC, "pcpx_curve_table.h"),
           os.path.join(INCLUDE, "pcpx.h")]
ARCH = "gfx950"
# -ffp-contract=off: the reference evaluates dx*dx+dy*dy+dz*dz without FMA; neighbour order and the
# eigen-solver restatement are only bit-comparable with it if the GPU does not fuse either.
# -fno-slp-vectorize: SLP packs the scalar f32 distance code into v_pk_* ops plus v_mov shuffles; packed
# f32 is not faster than scalar VALU on gfx950 and the shuffles cost ~4 % (measured, tools/ab_variants.py).
# -Wall -Werror=uninitialized: a self-initialised index (`u32 tid = tid;`) once reached the GPU as a wild address.
# -Werror=inline-asm: no reserved register (m0, ...) on an asm clobber list -- hipcc only warns that it "may not be preserved".
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=" + ARCH,
         "-Wall", "-Werror=uninitialized", "-Werror=inline-asm", "-I" + INCLUDE]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), tag=""):
    """tag/extra_flags build a tuning variant libpcpx_<tag>.so (tools/ab_variants.py); the default build
    has neither."""
    obj_dir = OBJ_DIR + ("_" + tag if tag else "")
    lib = LIB if not tag else os.path.join(HERE, "libpcpx_%s.so" % tag)
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    hipcc = _hipcc()
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
