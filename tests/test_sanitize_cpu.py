"""CPU sanitizer builds (SURVEY §5 "compile with -fsanitize=address on the host side"; GPU
ASan is not available on this pool).  The C oracle is compiled with ASan + UBSan and run
over the edge cases of the parity tests; a sanitizer report makes the program exit non-zero."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_oracle_under_address_and_undefined_sanitizers(tmp_path):
    exe = str(tmp_path / "oracle_asan")
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-fopenmp", "-Wall", "-Werror",
           os.path.join(ROOT, "tests", "sanitize", "oracle_driver.c"),
           os.path.join(ROOT, "oracle", "layer_oracle.c"), "-o", exe, "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="4")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.strip() == "ok"
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_library_host_side_under_address_and_undefined_sanitizers(tmp_path):
    """The HOST half of every .hip source of libipde_hip.so (--offload-host-only) with
    ASan + UBSan, linked with a C driver that walks the argument-checking / failure paths
    that run before any device work (tests/sanitize/abi_driver.c): context creation without
    (or with) a usable GPU, NULL contexts / plans / handles on every entry-point family."""
    from concurrent.futures import ThreadPoolExecutor
    from ipde_amd.build import SOURCES, CSRC
    flags = ["--offload-host-only", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=all", "-std=c++17", "-fPIC", "-w", "-I/opt/rocm/include"]

    def cc(src):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        r = subprocess.run([HIPCC, *flags, "-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        return obj
    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, [os.path.join(CSRC, s) for s in SOURCES]))
    drv = str(tmp_path / "abi_driver.o")
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-Wall", "-Werror", "-c",
                        os.path.join(ROOT, "tests", "sanitize", "abi_driver.c"), "-o", drv],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # host-only objects still reference their (absent) device code objects: give every
    # `__hip_fatbin_<hash>` a zero-filled stand-in — nothing is ever launched here
    nm = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True).stdout
    fat = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin_" in ln})
    stub = str(tmp_path / "fatbin_stubs.c")
    with open(stub, "w") as fh:
        fh.write("".join("char %s[128] = {0};\n" % name for name in fat))
    stub_o = str(tmp_path / "fatbin_stubs.o")
    subprocess.check_call(["gcc", "-c", stub, "-o", stub_o])
    exe = str(tmp_path / "abi_asan")
    r = subprocess.run([HIPCC, "-fsanitize=address,undefined", "-o", exe, drv, stub_o, *objs,
                        "-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # (leak detection off: the HIP runtime keeps process-lifetime allocations)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.strip().endswith("ok")
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
