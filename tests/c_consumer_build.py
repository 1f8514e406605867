"""Builds examples/c_abi_consumer.c with gcc (plain C: the header must be C, not C++) against include/gmpe.h, libgmpe.so and the HIP runtime."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contracts-marl-aam-corridors_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_c_consumer(outdir):
    exe = os.path.join(outdir, "c_abi_consumer")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           os.path.join(ROOT, "examples", "c_abi_consumer.c"), "-o", exe, "-L", PKG, "-lgmpe", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    subprocess.check_call(cmd)
    return exe
