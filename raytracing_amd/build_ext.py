"""Builds the HIP library in-tree: ``raytracing_amd/libraytracing_amd.so`` for gfx950.

hipcc cross-compiles without a GPU, so this runs anywhere the ROCm toolchain is installed.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libraytracing_amd.so")
SOURCES = ["scene.hip", "occlusion.hip", "solve.hip", "visibility.hip", "sharp_edges.hip", "fields.hip", "channel_stats.hip",
           "radiomap.hip", "reference_solver.hip", "table_format.hip"]
HEADERS = ["rt_common.h", "rt_candidates.h", "rt_fresnel.h", "rt_hybrd.h", os.path.join("..", "..", "include", "rt_pathsolve.h")]
# packed-fp32 instructions are switched off: the compiler forms them by itself from ordinary float code and some of
# their forms have given wrong results on this hardware (the host pass warns that it does not know the feature)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


FLAGS_STAMP = LIB + ".flags"


def _flags_line():
    return " ".join(FLAGS + os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split())


def _object_of(source):
    return os.path.join(HERE, "build", os.path.splitext(source)[0] + ".o")


def _newer_than(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_changed():
    try:
        with open(FLAGS_STAMP) as f:
            return f.read() != _flags_line()  # built with other flags (e.g. an instrumented build)
    except OSError:
        return True


def build(force=False, verbose=False):
    """One object file per source (compiled side by side, only the ones whose source or a header changed), then the
    link.  The sources share no device symbols, so no relocatable device code is needed."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DRT_OCC_STATS for the obstruction work counters
    force = force or _flags_changed()
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + extra
    todo = [s for s in SOURCES if force or _newer_than(_object_of(s), [os.path.join(CSRC, s)] + headers)]

    def compile_one(source):
        cmd = [hipcc] + compile_flags + ["-c", os.path.join(CSRC, source), "-o", _object_of(source)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        # the host pass does not know the packed-fp32 feature and says so once per file: harmless, not shown
        noise = "is not a recognized feature for this target"
        out = "\n".join(line for line in res.stdout.splitlines() if noise not in line)
        if out.strip():
            print(out)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {source}")

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), int(os.environ.get("MAX_JOBS", "8")))) as pool:
            list(pool.map(compile_one, todo))
    objects = [_object_of(s) for s in SOURCES]
    if todo or _newer_than(LIB, objects):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objects + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(FLAGS_STAMP, "w") as f:
            f.write(_flags_line())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
