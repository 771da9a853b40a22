#!/bin/bash
# Sanitizer runs of the native HOST code (CPU builds only; GPU sanitizers are not available on this pool).
#   ASan+UBSan and TSan builds of pangaea_amd/csrc/host.cpp (ingest incl. the threaded parser, packing, planning, TNF
#   columns, CSV writer, bin writer) and an ASan+UBSan build of the oracle, driven through ctypes.
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=$(mktemp -d)
CXXFLAGS="-O1 -g -std=c++17 -fPIC -shared -Iinclude -Ipangaea_amd/csrc"
g++ $CXXFLAGS -fsanitize=address,undefined -fno-sanitize-recover=undefined pangaea_amd/csrc/host.cpp -o $OUT/host_asan.so -lz -lpthread
g++ $CXXFLAGS -fsanitize=thread pangaea_amd/csrc/host.cpp -o $OUT/host_tsan.so -lz -lpthread
echo "== ASan + UBSan: host.cpp"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/sanitize_host.py $OUT/host_asan.so
echo "== TSan: host.cpp (threaded ingest)"
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" python3 tools/sanitize_host.py $OUT/host_tsan.so
echo "== ASan + UBSan: oracle"
make -s -C oracle asan
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 PG_ORACLE_LIB=$PWD/oracle/liboracle_asan.so \
    python3 -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider
rm -rf $OUT oracle/liboracle_asan.so
echo "sanitizers: clean"
