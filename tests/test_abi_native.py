"""The C ABI from a plain C host (tests/abi/abi_client.c): no Python, no torch on the call path.  The CPU test
checks that the client compiles and links against include/uavx.h + libuavx.so with gcc; the GPU test runs it
(device buffers from hipMalloc, launches on its own hipStream) against the oracle's C library."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym_uav_collision_avoidance_amd", "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path, oracle_mod, name="abi_client"):
    from gym_uav_collision_avoidance_amd import _lib
    _lib.build()                       # no-op when libuavx.so is current
    oracle_dir = os.path.dirname(oracle_mod.build())
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-std=gnu11", "-Wall", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "abi", name + ".c"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROCM, "include"),
           "-L" + CSRC, "-luavx", "-L" + oracle_dir, "-luavx_oracle", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + CSRC, "-Wl,-rpath," + oracle_dir, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
@pytest.mark.parametrize("name", ["abi_client", "abi_client_ext"])
def test_c_client_compiles_and_links(tmp_path, oracle_mod, name):
    exe = _build(tmp_path, oracle_mod, name)
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("envs,agents,steps", [(512, 4, 400), (96, 9, 300)])
def test_c_client_parity_on_device(tmp_path, oracle_mod, envs, agents, steps):
    exe = _build(tmp_path, oracle_mod)
    out = subprocess.run([exe, str(envs), str(agents), str(steps)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "mismatches 0" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("envs,learners,bodies,steps", [(256, 8, 16, 160), (300, 3, 5, 120), (64, 24, 0, 90)])
def test_c_client_extension_parity_on_device(tmp_path, oracle_mod, envs, learners, bodies, steps):
    """ABI version 2 from plain C: scripted bodies, curriculum levels, uavx_step_ex with ended / truncated (extension:
    checked against the oracle's restatement, no reference counterpart)."""
    exe = _build(tmp_path, oracle_mod, "abi_client_ext")
    out = subprocess.run([exe, str(envs), str(learners), str(bodies), str(steps)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "mismatches 0" in out.stdout
