"""Build the HIP extension in-tree: python -m indexed_bzip2_amd.build

Compiles csrc/*.hip and csrc/*.cpp with hipcc for gfx950 only into indexed_bzip2_amd/libmi355x_bz2.so
and csrc/ibzip2_cli.cpp into the command line tool indexed_bzip2_amd/ibzip2-mi355x
(both git-ignored; they travel to the GPU box with the gpurun snapshot).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmi355x_bz2.so")


CLI_SRC = os.path.join(CSRC, "ibzip2_cli.cpp")
CLI_OUT = os.path.join(HERE, "ibzip2-mi355x")


def sources():
    files = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))
    return sorted(f for f in files if f != CLI_SRC)


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(CLI_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(CLI_OUT))
    deps = (sources() + [CLI_SRC] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp"))
            + glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread",
           "-Wall", "-Wno-unused-function", "-o", OUT] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # command line tool over the C ABI (finds the library next to itself)
    cli = [hipcc, "-O2", "-std=c++17", "-Wall", "-o", CLI_OUT, CLI_SRC, "-L" + HERE, "-lmi355x_bz2", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cli))
    subprocess.run(cli, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
