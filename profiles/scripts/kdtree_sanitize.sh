# Host-only sanitizer runs of the point-order routine (no GPU): AddressSanitizer + UBSan, then
# ThreadSanitizer, over 40 clouds (1 to 200 k points, duplicates, constant, clustered), each ordered on one
# thread and on 2 .. 8 threads of the worker pool and compared.
# usage (repo root): bash profiles/scripts/kdtree_sanitize.sh
set -e
tmp=$(mktemp -d)
cat > $tmp/stub.cpp <<'CPP'
#include <cstdarg>
#include <cstdio>
namespace mvf { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
CPP
for san in address,undefined thread; do
  g++ -O1 -g -std=c++17 -pthread -fsanitize=$san -fno-omit-frame-pointer profiles/microbench/kdtree_sanitize.cpp $tmp/stub.cpp \
      multimodal-fusion_amd/csrc/mvf_kdtree.cpp -o $tmp/drv
  echo "== -fsanitize=$san"; $tmp/drv
done
rm -rf $tmp
