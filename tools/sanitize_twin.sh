#!/bin/bash
# TEST-ONLY: AddressSanitizer + UBSan build of the host twin (CPU; GPU sanitizers are not available on this pool) and a
# run of the parity cases through it.  ~20 min of compile time on 8 cores.  usage: tools/sanitize_twin.sh [outdir]
# TSAN=1 tools/sanitize_twin.sh [outdir]: ThreadSanitizer instead, and only the cases in which several host threads meet --
# the issuer's contexts shared by threads, one context driven by several threads, the pool's member threads and its lists in
# flight, completion-order retire (round 5: the data races of issuer.hpp the previous review listed).
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ -n "$TSAN" ]; then
  OUT=${1:-$ROOT/bbs_sign_amd/build/tsan}; mkdir -p $OUT
  if [ -z "$SKIP_BUILD" ] || [ ! -f $OUT/libbbs_hosttwin_tsan_TESTONLY.so ]; then
    ls $ROOT/bbs_sign_amd/csrc/*.hip | xargs -P ${JOBS:-8} -I{} sh -c "hipcc -O1 -g --offload-host-only -x hip -DBBS_HOST_TWIN -DBBS_CHECK_BOUNDS -fPIC -fsanitize=thread -fno-omit-frame-pointer -c {} -o $OUT/\$(basename {} .hip).o"
    hipcc -shared -fPIC --offload-host-only -fsanitize=thread -shared-libsan $OUT/*.o -o $OUT/libbbs_hosttwin_tsan_TESTONLY.so
  fi
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
  cat > $OUT/run_tsan.py <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import parity_cases as pc
lib = "$OUT/libbbs_hosttwin_tsan_TESTONLY.so"
for curve in ("bls12_381", "bn254"):
    pc.check_issuer_threads(curve, lib, threads=3, rounds=2)
    pc.check_issuer_budget(curve, lib)
    pc.check_submit(curve, lib)
    print(curve, "ok", flush=True)
pc.check_threads(lib, threads=3, rounds=2, n=5)
pc.check_pool(lib, devices=(0, 0, 0), per_curve=13, max_batch=3)
print("thread sanitizer run ok")
PY
  set +e
  TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 LD_PRELOAD=$RT python3 $OUT/run_tsan.py > $OUT/run.log 2>&1
  rc=$?
  grep -E "WARNING: ThreadSanitizer|ok" $OUT/run.log | sort | uniq -c
  if [ $rc -ne 0 ] || ! grep -q "thread sanitizer run ok" $OUT/run.log || grep -q "WARNING: ThreadSanitizer" $OUT/run.log; then echo "THREAD SANITIZER RUN FAILED OR REPORTED (rc=$rc)"; grep -A25 "WARNING: ThreadSanitizer" $OUT/run.log | head -120; exit 1; fi
  [ -n "$KEEP_SAN" ] || rm -f $OUT/*.o $OUT/libbbs_hosttwin_tsan_TESTONLY.so
  exit 0
fi
OUT=${1:-$ROOT/bbs_sign_amd/build/san}; mkdir -p $OUT
pids=()
# SKIP_BUILD=1 with an instrumented library present: neither compile nor link (the objects may be gone)
if [ -z "$SKIP_BUILD" ] || [ ! -f $OUT/libbbs_hosttwin_san_TESTONLY.so ]; then
  for tu in $ROOT/bbs_sign_amd/csrc/*.hip; do
    hipcc -O1 -g --offload-host-only -x hip -DBBS_HOST_TWIN -DBBS_CHECK_BOUNDS -fPIC -fsanitize=address,undefined \
          -fno-omit-frame-pointer -fno-sanitize=vptr -c $tu -o $OUT/$(basename $tu .hip).o &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  hipcc -shared -fPIC --offload-host-only -fsanitize=address,undefined -shared-libsan $OUT/*.o -o $OUT/libbbs_hosttwin_san_TESTONLY.so.tmp
  mv $OUT/libbbs_hosttwin_san_TESTONLY.so.tmp $OUT/libbbs_hosttwin_san_TESTONLY.so
fi
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cat > $OUT/run_san.py <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import parity_cases as pc
lib = "$OUT/libbbs_hosttwin_san_TESTONLY.so"
pc.check_kat_vectors(lib)
for curve in ("bls12_381", "bn254"):
    pc.check_golden(curve, lib, max_L=5)
    pc.check_random_batch(curve, lib, n=6, L=3, seed=3)
    pc.check_error_semantics(curve, lib)
    pc.check_batch_verification(curve, lib)
    pc.check_points_in_subgroup(curve, lib)
    pc.check_pippenger(curve, lib, n=24)
    pc.check_empty_batches(curve, lib)
    pc.check_fail_closed(curve, lib)
    pc.check_submit(curve, lib)
    pc.check_latency_mode(curve, lib)
    pc.check_window_widths(curve, lib, widths=(5, 13))
    pc.check_large_shapes(curve, lib, L=40, n=2)
    pc.check_proof_verify_octets(curve, lib)
    pc.check_proof_verify_octets(curve, lib, seed=63, disclose_all_3=True)
    pc.check_verify_octets(curve, lib)
    pc.check_octets_out(curve, lib)
    pc.check_proof_verify_wire(curve, lib)
    pc.check_sign_verify_wire(curve, lib)
    pc.check_fixed_base_tree(curve, lib, n_pv=6)
    pc.check_issuer_mixed_lengths(curve, lib)
    pc.check_issuer_budget(curve, lib)
    pc.check_proof_gen_unusual_points(curve, lib)
    pc.check_issuer_threads(curve, lib, threads=3, rounds=2)
    pc.check_big_batch(curve, lib, n=66, L=4, R=2, window_bits=4)
    print(curve, "ok", flush=True)
pc.check_threads(lib, threads=3, rounds=2, n=5)
print("sanitizer run ok")
PY
# SKIP_BUILD=1 tools/sanitize_twin.sh re-runs the cases against an existing instrumented library
set +e
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$RT python3 $OUT/run_san.py > $OUT/run.log 2>&1
rc=$?
grep -E "runtime error|AddressSanitizer|ok" $OUT/run.log | sort | uniq -c
if [ $rc -ne 0 ] || ! grep -q "sanitizer run ok" $OUT/run.log; then echo "SANITIZER RUN FAILED (rc=$rc)"; tail -20 $OUT/run.log; exit 1; fi
# the instrumented objects are ~400 MB and would ride along with every gpurun snapshot: remove them unless asked to keep
[ -n "$KEEP_SAN" ] || rm -f $OUT/*.o $OUT/libbbs_hosttwin_san_TESTONLY.so
