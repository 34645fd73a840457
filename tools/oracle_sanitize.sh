#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UBSan: builds oracle/*.c with -fsanitize=address,undefined into /tmp and runs the
# CPU tests that call it.  (GPU AddressSanitizer is not available on this pool: sanitizers run on the CPU build only.)   bash tools/oracle_sanitize.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=/tmp/stvo_san; mkdir -p $D
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -march=x86-64-v3 -ffp-contract=off -fPIC -std=gnu99 -shared \
    -o $D/liboracle.so $R/oracle/stvo_oracle.c $R/oracle/stvo_orb_oracle.c $R/oracle/stvo_lbd_oracle.c $R/oracle/stvo_lsd_oracle.c -I$R/oracle -lm
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  STVO_ORACLE_SO=$D/liboracle.so timeout 1500 python -m pytest tests/ -x -q -m "not gpu" "$@"
