"""Builds mbb_emcee_amd/libmbb_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc.

    python -m mbb_emcee_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is built in-tree so that it
travels with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mbb_hip.hip")
SRC_HOST = os.path.join(HERE, "csrc", "mbb_host_tables.cpp")      # host-only table builders
SRC_FLOW = os.path.join(HERE, "csrc", "mbb_flow.hip")             # the one-launch sampler kernel, own flags
SRC_REG = os.path.join(HERE, "csrc", "mbb_registry.cpp")          # host-only: which processes are on which GPU
# Both device translation units: no contraction of a product and a sum the source keeps apart.  The
# sampler forms (k_lnlike SMODE 1/2/5/6, k_flowm) are held to one another bit for bit, and the shared
# arithmetic is inlined into each of them: with the compiler free to contract, whether a given a*b+c
# rounds once or twice depends on the code around it.  Every intended fused multiply-add is an explicit
# fma() in the source (tools/fma_audit.py lists the kernels whose fp64 fma/mul/add counts differ between
# the two settings: none in the sample loop; profiles/r03/fma_audit.txt).
DEVICE_FLAGS = ["-ffp-contract=off"]
FLOW_FLAGS = ["-mllvm", "-sink-insts-to-avoid-spills", "-mllvm", "-disable-machine-licm"]
DEPS = [SRC, SRC_HOST, SRC_FLOW, SRC_REG, os.path.join(HERE, "csrc", "mbb_host_tables.h"),
        os.path.join(HERE, "csrc", "mbb_registry.h"),
        os.path.join(HERE, "csrc", "mbb_exp2_tab.inc"),
        os.path.join(HERE, "csrc", "mbb_walker_consts.inc"),
        os.path.join(HERE, "csrc", "mbb_walker_penalties.inc"),
        os.path.join(HERE, "csrc", "mbb_flow_index.h"),
        os.path.join(HERE, "csrc", "mbb_device.hip.h"),
        os.path.join(HERE, "csrc", "mbb_math.hip.h"),
        os.path.join(HERE, "csrc", "mbb_kernels.hip.h"),
        os.path.join(HERE, "csrc", "mbb_flowm.hip.h"),
        os.path.join(HERE, "csrc", "mbb_flowa.hip.h"),
        os.path.join(HERE, "csrc", "mbb_serve.hip.h"),
        os.path.join(os.path.dirname(HERE), "include", "mbb_hip.h")]
LIB = os.path.join(HERE, "libmbb_hip.so")
ARCH = "gfx950"


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build the MI355X likelihood library")


def flags_tag(extra_flags=()):
    """What a library was built with, as written beside it (`<lib>.flags`)."""
    return " ".join(DEVICE_FLAGS + ["|"] + FLOW_FLAGS + ["|"] + list(extra_flags))


def needs_build(lib=None, extra_flags=()):
    """Missing, older than a source, or built with other flags than asked for now (an A/B or -DMBB_STAMPS
    library left by an earlier run is not handed back as up to date)."""
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    if any(os.path.getmtime(d) > t for d in DEPS):
        return True
    try:
        return open(lib + ".flags").read() != flags_tag(extra_flags)
    except OSError:
        return bool(extra_flags)         # (a library without the note is taken as a default build)


def build(force=False, verbose=False, extra_flags=(), out=None, obj_tag=""):
    """Builds `out` (default: the in-tree library) when it is missing or older than a source and
    returns its path.  Processes that import at the same time (the ranks of a sharded test) take a
    file lock, compile into an object directory of their own tag, and the finished library replaces
    the old one in one rename, so nobody ever maps a half-written file."""
    import fcntl
    target = out or LIB
    if not force and not needs_build(target, extra_flags):
        return target
    objdir = os.path.join(HERE, "csrc", "_obj" + obj_tag)
    os.makedirs(objdir, exist_ok=True)
    # the lock belongs to the target: two builds of one library exclude each other whatever their object tags
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build(target, extra_flags):       # another process built it while we waited
            return target
        # four objects (the two device translation units in parallel), then one link
        common = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + DEVICE_FLAGS + list(extra_flags)
        if verbose:
            common.append("-Rpass-analysis=kernel-resource-usage")
        jobs = [(SRC, os.path.join(objdir, "mbb_hip.o"), []),
                (SRC_FLOW, os.path.join(objdir, "mbb_flow.o"), FLOW_FLAGS),
                (SRC_HOST, os.path.join(objdir, "mbb_host_tables.o"), []),
                (SRC_REG, os.path.join(objdir, "mbb_registry.o"), [])]
        procs = []
        for src, obj, flags in jobs:
            cmd = common + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        tmp = "%s.tmp.%d" % (target, os.getpid())
        subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] +
                              [obj for _, obj, _ in jobs] + ["-ldl"])
        # the library first, then the note of its flags: a process that dies in between leaves a NEW library beside an OLD
        # note -- at worst a mismatch, i.e. one more build, never an old library passed off as built with the new flags
        os.replace(tmp, target)
        with open(target + ".flags.tmp.%d" % os.getpid(), "w") as f:
            f.write(flags_tag(extra_flags))
        os.replace(f.name, target + ".flags")
    return target


SRC_FAST = os.path.join(HERE, "csrc", "mbb_fastcall.c")
FAST = os.path.join(HERE, "_mbbfast.so")


def fastcall_is_current(abi=None):
    """Is the built `_mbbfast` extension this interpreter's and this numpy's, and not older than its source?"""
    import sysconfig
    try:
        if abi is None:
            import numpy
            abi = "%s | numpy %s" % (sysconfig.get_config_var("EXT_SUFFIX") or sys.version, numpy.__version__)
        return (os.path.exists(FAST) and os.path.getmtime(FAST) >= os.path.getmtime(SRC_FAST)
                and open(FAST + ".abi").read() == abi)
    except Exception:           # noqa
        return False


def build_fastcall(force=False):
    """Builds the CPython extension `_mbbfast` (csrc/mbb_fastcall.c: likelihood.__call__'s boundary call as one C-level
    callable) with the host compiler and returns its path, or None when that is not possible here (no compiler, no
    Python or numpy headers): the likelihood then makes the same call through numpy and ctypes."""
    import sysconfig
    try:
        import numpy
        # the extension is good for the interpreter and the numpy C-API it was compiled against: both are noted beside it
        # (`_mbbfast.so.abi`) and another Python or numpy in the same checkout builds it anew instead of loading it
        abi = "%s | numpy %s" % (sysconfig.get_config_var("EXT_SUFFIX") or sys.version, numpy.__version__)
        if not force and fastcall_is_current(abi):
            return FAST
        cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
        if not cc:
            return None
        tmp = "%s.tmp.%d" % (FAST, os.getpid())
        subprocess.check_call([cc, "-O2", "-fPIC", "-shared", "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(),
                               SRC_FAST, "-o", tmp], stderr=subprocess.DEVNULL)
        os.replace(tmp, FAST)
        with open(FAST + ".abi.tmp.%d" % os.getpid(), "w") as f:
            f.write(abi)
        os.replace(f.name, FAST + ".abi")
        return FAST
    except Exception:           # noqa -- an optional accelerator of the host glue, never a reason to fail a build
        return None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_fastcall(force="--force" in sys.argv))
