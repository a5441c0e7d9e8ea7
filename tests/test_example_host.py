"""The compiled host of examples/ (C++ on the C ABI, no Python in the process): builds against the header and the library; without a GPU it
must fail loudly at world creation (no CPU fallback), on a GPU it runs the closed loop and checks its own result."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "examples", "closed_loop_demo")


def build():
    subprocess.run(["make", "-C", os.path.join(REPO, "examples")], check=True, capture_output=True)
    assert os.path.exists(EXE)


def test_example_builds_and_refuses_to_run_without_a_device():
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([EXE, "3"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_example_runs_the_closed_loop_on_the_device():
    build()
    r = subprocess.run([EXE, "12", "10", "12", "60"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "DEMO_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_host_shapes_example_capsules_from_a_compiled_host():
    """examples/host_shapes_demo.cpp: capsules that exist only in the host's two callbacks next to a device box stack, in the library's closed loop."""
    build()
    exe = os.path.join(REPO, "examples", "host_shapes_demo")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "12", "10", "12", "400", "150"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "HOST_SHAPES_DEMO_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_collision_hooks_example_a_belt_and_ghosts_from_a_compiled_host():
    """examples/collision_hooks_demo.cpp: CollisionHooks::filter_pairs / modify_contacts as the callbacks of a compiled host (a conveyor belt through tangent_velocity,
    boxes that pass through each other through the pair filter) next to an unhooked pile, in the library's closed loop."""
    build()
    exe = os.path.join(REPO, "examples", "collision_hooks_demo")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "12", "10", "12", "8", "16", "150"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "HOOKS_DEMO_OK" in r.stdout, r.stdout + r.stderr
