"""Build the kernel library of another git revision next to the current one (for same-process A/B on the GPU box, tools/lib_ab.py):

    python tools/build_side.py <git-rev> <name>        ->  ullsam_amd/lib/libullsam_hip_<name>.so

The sources of <git-rev> (ullsam_amd/csrc, include/) are checked out into a scratch directory and compiled with ullsam_amd/build.py's flags.
The side library is git-ignored but travels to the GPU box with the snapshot.  A developer tool; nothing in the product loads it."""
import concurrent.futures as cf
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ullsam_amd import build as B  # noqa: E402


def main():
    rev, name = sys.argv[1], sys.argv[2]
    out = os.path.join(B.LIBDIR, f"libullsam_hip_{name}.so")
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(f"git -C {ROOT} archive {rev} ullsam_amd/csrc include | tar -x -C {td}", shell=True, check=True)
        csrc, inc = os.path.join(td, "ullsam_amd", "csrc"), os.path.join(td, "include")
        srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))

        def cc(src):
            obj = os.path.join(td, src.replace(".hip", ".o"))
            r = subprocess.run([B.HIPCC, *B.FLAGS[:-1], inc, "-c", os.path.join(csrc, src), "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-3000:])
            return obj

        with cf.ThreadPoolExecutor(max_workers=6) as ex:
            objs = list(ex.map(cc, srcs))
        subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    print(out)


if __name__ == "__main__":
    main()
