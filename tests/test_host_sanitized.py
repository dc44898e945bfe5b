"""The host logic of the library -- SpGEMM and polynomial rows of the set-up
(`hostcsr.hpp`), the pair-format builder (`pair_host.hpp`), partition and halo
index lists (`halo_host.hpp`) -- compiled WITHOUT HIP under AddressSanitizer +
UndefinedBehaviorSanitizer and driven over a small saddle system, whole and in
row blocks of 1..4 ranks (`tests/host_sanitize.cpp`).  CPU only: GPU sanitizer
runs are not available on the pool, and the round-3 memory fault was a
host-side lifetime bug."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_logic_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / 'host_sanitize')
    build = subprocess.run(
        ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined',
         '-fno-sanitize-recover=all', '-Wall', '-Wextra', '-Werror',
         '-I' + os.path.join(ROOT, 'include'),
         os.path.join(HERE, 'host_sanitize.cpp'), '-o', exe, '-lpthread'],
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert build.returncode == 0, build.stdout.decode()[-4000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0',
               UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([exe], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600)
    out = run.stdout.decode()
    assert run.returncode == 0, out[-4000:]
    assert 'all checks passed' in out
    assert 'runtime error' not in out and 'AddressSanitizer' not in out, out
