"""A plain-C program against include/dsea.h + libdsea.so (examples/c_abi/lanczos_stencil.c): built with gcc, run on the GPU --
the drop-in boundary exercised without Python or torch in the process."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [64, 300, 2000])
def test_plain_c_consumer_of_the_c_abi(tmp_path, n):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / HIP headers on this box")
    lib = os.path.join(ROOT, "dominantsparseeigenad_amd", "csrc")
    assert os.path.exists(os.path.join(lib, "libdsea.so")), "libdsea.so missing: run __graft_entry__.build()"
    exe = str(tmp_path / "lanczos_stencil")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi", "lanczos_stencil.c"),
                            "-L" + lib, "-ldsea", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib,
                            "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=120)
    sys.stdout.write(run.stdout)
    assert run.returncode == 0 and run.stdout.strip().endswith("PASS"), run.stdout + run.stderr


@pytest.mark.parametrize("n", [300, 2000])
def test_plain_c_consumer_of_the_row_partitioned_entry_points(tmp_path, n):
    """examples/c_abi/partitioned_stencil.c: dsea_comm_create_callbacks (the caller's own transport) and dsea_comm_unique_id /
    dsea_comm_init_rank (library-created communicators over the process's RCCL), dsea_pop_create_stencil3,
    dsea_pop_lanczos_run / _status, dsea_pop_cg_run, dsea_pop_matvec from plain C -- spectrum ends against the closed form,
    CG residual formed through the library's own mat-vec."""
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / HIP headers on this box")
    lib = os.path.join(ROOT, "dominantsparseeigenad_amd", "csrc")
    exe = str(tmp_path / "partitioned_stencil")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi", "partitioned_stencil.c"),
                            "-L" + lib, "-ldsea", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib,
                            "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))   # (librccl.so for form (b))
    run = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=180, env=env)
    sys.stdout.write(run.stdout)
    assert run.returncode == 0 and run.stdout.strip().endswith("PASS"), run.stdout + run.stderr
    assert "caller-supplied collectives" in run.stdout
