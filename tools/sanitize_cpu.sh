#!/bin/bash
# CPU-only sanitizer pass (GPU ASan is not available on the pool): the plain-C oracle and the C++ host mirror under
# AddressSanitizer + UndefinedBehaviorSanitizer.  Run from the repo root: bash tools/sanitize_cpu.sh
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/oracle" libfrw_oracle_asan.so >/dev/null
cat > /tmp/frw_asan_run.py <<'PY'
import sys, numpy as np, random
sys.path.insert(0, sys.argv[1] + "/tests"); sys.path.insert(0, sys.argv[1])
import frw_testlib as T
o = T.Oracle(sys.argv[1] + "/oracle/libfrw_oracle_asan.so")
for logn in (9, 10):
    sig, pk, hm, _ = T.random_triple(logn, random.Random(1))
    S, P, H = (np.stack([a] * 3) for a in (sig, pk, hm))
    for enc in (0, 1):
        o.witness_ntt_verify(logn, S, P, H, enc, threads=3)
        o.witness_dual_ntt_verify(logn, S, P, H, enc)
    o.ntt_modq(logn, np.stack([pk] * 2), 1)
    o.ntt_clear(logn, pk); o.ntt_clear(logn, pk, inverse=True)
    w, i, _ = o.witness_ntt_verify(logn, sig, pk, hm, 1)
    for name, arr in (("sig", sig), ("pk", pk), ("hm", hm), ("wit", w), ("inst", i)):
        arr.tofile("/tmp/frw_asan_%d_%s.bin" % (logn, name))
print("oracle: clean")
PY
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python /tmp/frw_asan_run.py "$ROOT"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/frw_mirror_asan \
    "$ROOT/tests/cpp/test_host_mirror.cpp" -L"$ROOT/falcon-r1cs_amd" -lfrw -Wl,-rpath,"$ROOT/falcon-r1cs_amd"
ASAN_OPTIONS=detect_leaks=0 /tmp/frw_mirror_asan structure | tail -1
for logn in 9 10; do
  ASAN_OPTIONS=detect_leaks=0 /tmp/frw_mirror_asan check $logn /tmp/frw_asan_${logn}_sig.bin /tmp/frw_asan_${logn}_pk.bin \
      /tmp/frw_asan_${logn}_hm.bin /tmp/frw_asan_${logn}_wit.bin /tmp/frw_asan_${logn}_inst.bin
done
echo "host mirror: clean"
# the field / curve headers of the prover (frw_fq29.h incl. the Euclidean inversion, frw_quad.h's level programmes) through their host
# harness: the sanitized library takes the place of the test's own for one run of tests/test_fq29_host.py
# (a path of its own, named to the test through FRW_TEST_FQ29_SO, and removed whatever happens: a sanitized library left where the
# ordinary fixture looks would be newer than the sources and fail to load without the preload)
SO=$(mktemp /tmp/libtest_fq29_asan.XXXXXX.so)
trap 'rm -f "$SO"' EXIT
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-unknown-pragmas -fno-strict-aliasing \
    -I "$ROOT/tests/cpp/hip_host" -I "$ROOT/falcon-r1cs_amd/csrc" -o "$SO" "$ROOT/tests/cpp/test_fq29.cpp"
(cd "$ROOT" && FRW_TEST_FQ29_SO="$SO" ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_fq29_host.py -x -q 2>&1 | tail -1)
echo "fq29 / quad headers: clean"
