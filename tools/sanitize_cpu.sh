#!/bin/bash
# AddressSanitizer + UBSan over the CPU builds (GPU sanitizers are not available on the pool): the device field / curve
# headers compiled for the host with bound tracking (csrc/host_check.cpp) and the C oracle, each under its pytest file.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
g++ -O1 -g -std=c++17 -DHM_BOUNDS -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -o /tmp/libhm_hostcheck_asan.so halo2-experiments_amd/csrc/host_check.cpp
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -pthread -o /tmp/libcpu_ref_asan.so oracle/cpu_ref.c -lm
export ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so)"
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from halo2_experiments_amd import _lib
from oracle import cpu_ref
_lib.HOSTCHECK_PATH = '/tmp/libhm_hostcheck_asan.so'
cpu_ref._LIB = '/tmp/libcpu_ref_asan.so'
import pytest
sys.exit(pytest.main(['-x', '-q', 'tests/test_ff29_host.py', 'tests/test_oracle.py', '-p', 'no:cacheprovider']))
PY
