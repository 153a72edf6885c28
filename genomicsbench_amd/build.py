"""In-tree build of every native artefact (called by __graft_entry__.build())."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "genomicsbench_amd")


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def build_libgbx(jobs=4):
    """HIP kernels + C-ABI -> genomicsbench_amd/libgbx.so (hipcc --offload-arch=gfx950)."""
    _run(["make", "-j%d" % jobs], cwd=os.path.join(PKG, "csrc"))


def build_datagen():
    src = os.path.join(PKG, "datagen", "datagen.c")
    out = os.path.join(PKG, "libgbx_datagen.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        _run(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", src, "-o", out, "-lm"])


def build_oracle():
    """CPU restatements (test infrastructure) and, when /root/reference exists, the reference shims."""
    _run(["make"], cwd=os.path.join(ROOT, "oracle"))
    _run(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])


def build_drivers():
    d = os.path.join(PKG, "csrc", "drivers")
    if os.path.exists(os.path.join(d, "Makefile")):
        _run(["make"], cwd=d)


def build_all():
    build_libgbx()
    build_datagen()
    build_drivers()
    build_oracle()


if __name__ == "__main__":
    build_all()
    sys.exit(0)
