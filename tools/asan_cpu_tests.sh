#!/bin/bash
# CPU-only sanitizer pass: libglrt_host.so, libglrt.so and the oracle built with -fsanitize=address,undefined, swapped
# in for the run of `pytest -m "not gpu"`, then the regular builds are restored.  The device library (hipcc) is not instrumented: GPU
# AddressSanitizer is not available on this pool.  Usage: tools/asan_cpu_tests.sh [log file]
set -u
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/opengl-raytracer_amd
LOG=${1:-$ROOT/profiles/r03_sanitizer.txt}
BK=$(mktemp -d)
cp $PKG/lib/libglrt_host.so $PKG/lib/libglrt.so $ROOT/oracle/_ref/libpt_oracle.so $BK/
# the regular builds come back whatever happens below (a failed compile, an interrupt): instrumented libraries left in the tree do not load without LD_PRELOAD
restore() {
  if [ -d "$BK" ]; then
    cp $BK/libglrt_host.so $BK/libglrt.so $PKG/lib/
    cp $BK/libpt_oracle.so $ROOT/oracle/_ref/
    rm -rf $BK
  fi
}
trap restore EXIT
trap 'exit 130' INT TERM
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
cd $PKG
g++ -O1 -std=c++17 -fPIC -Wall -Wextra -I$ROOT/include $SAN -shared -o lib/libglrt_host.so host/bvh.cpp host/camera.cpp || exit 1
g++ -O1 -std=c++17 -fPIC -Wall -Wextra -I$ROOT/include $SAN -shared -o lib/libglrt.so host/scene.cpp host/window.cpp -Llib -lglrtx -lglrt_host -Wl,-rpath,'$ORIGIN' || exit 1
gcc -O1 -fPIC -shared -Wall -Wextra -ffp-contract=off -fno-fast-math -fopenmp $SAN -o $ROOT/oracle/_ref/libpt_oracle.so $ROOT/oracle/pt_oracle.c -lm || exit 1
cd $ROOT
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
{
  echo "sanitizer pass: libglrt_host.so, libglrt.so, oracle/_ref/libpt_oracle.so built with $SAN (gcc $(gcc -dumpversion)); pytest -m 'not gpu'"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 \
    timeout 1800 python -m pytest tests -q -m "not gpu" -x -p no:cacheprovider > $BK/pytest.log 2>&1
  RC=$?   # pytest's own status, not the status of a filter behind it
  grep -v "^\[INFO\]" $BK/pytest.log | tail -40
  # UBSan reports and continues (halt_on_error=0): a report is a failure of this script all the same
  if grep -q "runtime error:" $BK/pytest.log; then echo "UBSan reported runtime errors:"; grep "runtime error:" $BK/pytest.log | sort | uniq -c | head -20; RC=1; fi
  echo "exit status $RC"
  exit $RC
} > $LOG 2>&1
RC=$?
tail -15 $LOG
exit $RC
