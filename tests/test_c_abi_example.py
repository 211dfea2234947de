"""examples/c_abi_demo.c: the C ABI from plain C99 (gcc, HIP runtime for the buffers, no Python / torch in the process).
CPU: it compiles and links against libelliot_hip.so.  GPU: it runs -- a few BPR-MF steps, masked top-5, host recomputation."""
import os
import shutil
import subprocess

import pytest

from elliot_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out, source="c_abi_demo.c", extra=()):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    csrc = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", *extra, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", source), "-L" + csrc, "-lelliot_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def test_plain_c_host_program_compiles_and_links(tmp_path):
    exe = build(str(tmp_path / "c_abi_demo"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_plain_c_host_program_runs(tmp_path):
    exe = build(str(tmp_path / "c_abi_demo"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "identical to the host recomputation" in r.stdout and "loss/triplet" in r.stdout
    losses = [float(l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("step")]
    assert len(losses) == 5 and losses[-1] < losses[0]


# ---- examples/c_abi_multigpu.c: one process per GPU, RCCL through el_comm_* from plain C ------------------------
def test_c_multi_gpu_host_program_compiles_and_links(tmp_path):
    exe = build(str(tmp_path / "c_abi_multigpu"), "c_abi_multigpu.c", ("-D_POSIX_C_SOURCE=200809L",))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_multi_gpu_host_program_runs_one_rank_per_gpu(tmp_path):
    """On the one-GPU box: world = 1 (RCCL refuses two ranks on one device); on a multi-GPU node the default is one rank per
    visible GPU -- the same binary."""
    exe = build(str(tmp_path / "c_abi_multigpu"), "c_abi_multigpu.c", ("-D_POSIX_C_SOURCE=200809L",))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "== single shard (indices and score bits)" in r.stdout and "all-reduce" in r.stdout
