#!/bin/bash
# tools/sanitize.sh -- AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on the pool): the
# oracle under its own tests, and the C++ host layer (parsers, Parquet reader, report) under the host tests.  Run at the root.
set -e
T=$(mktemp -d -p gpurun_out 2>/dev/null || mktemp -d)
PA=$(python3 -c "import pyarrow, os; print(os.path.dirname(pyarrow.__file__))")
gcc -O1 -g -fPIC -std=c11 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/libslacken_oracle.so oracle/slacken_oracle.c -lm
g++ -O1 -g -std=c++20 -DSLK_HAVE_PARQUET -I$PA/include -fsanitize=address,undefined -fno-omit-frame-pointer -c -o $T/pq.o slacken_amd/host/parquet_source.cpp
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o $T/slacken-amd slacken_amd/host/slacken_cli.cpp $T/pq.o \
  -L$PA -l:$(basename $(ls $PA/libparquet.so.[0-9]* | head -1)) -l:$(basename $(ls $PA/libarrow.so.[0-9]* | head -1)) -Wl,-rpath,$PA \
  -Lslacken_amd/lib -lslacken_amd -lz -ldl -lpthread -Wl,-rpath,$PWD/slacken_amd/lib
cp oracle/libslacken_oracle.so $T/oracle.orig; cp slacken_amd/bin/slacken-amd $T/cli.orig
trap 'cp $T/oracle.orig oracle/libslacken_oracle.so; cp $T/cli.orig slacken_amd/bin/slacken-amd; rm -rf $T' EXIT
cp $T/libslacken_oracle.so oracle/libslacken_oracle.so; cp $T/slacken-amd slacken_amd/bin/slacken-amd
export ASAN_OPTIONS=detect_leaks=0
LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_oracle_kat.py tests/test_oracle_props.py tests/test_oracle_lca.py tests/test_oracle_build.py tests/test_golden.py tests/test_config1.py -x -q -m "not gpu"
python -m pytest tests/test_host_cli.py tests/test_host_classify_gpu.py tests/test_pargz.py tests/test_parbz2.py -x -q -m "not gpu"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_title_grouping.py -x -q   # (loads the instrumented oracle too)
SLK_IO_CHUNK=3 python -m pytest tests/test_host_cli.py -x -q
# ThreadSanitizer over the threaded readers (segment parser, per-file inflate threads, batch prefetcher, recyclers)
g++ -O1 -g -std=c++17 -fsanitize=thread -o $T/slacken-amd-tsan slacken_amd/host/slacken_cli.cpp slacken_amd/bin/parquet_source.o \
  -L$PA -l:$(basename $(ls $PA/libparquet.so.[0-9]* | head -1)) -l:$(basename $(ls $PA/libarrow.so.[0-9]* | head -1)) -Wl,-rpath,$PA \
  -Lslacken_amd/lib -lslacken_amd -lz -ldl -lpthread -Wl,-rpath,$PWD/slacken_amd/lib
cp $T/slacken-amd-tsan slacken_amd/bin/slacken-amd
cp $T/oracle.orig oracle/libslacken_oracle.so
TSAN_OPTIONS=halt_on_error=1 SLK_PARSE_THREADS=6 python -m pytest tests/test_host_cli.py tests/test_title_grouping.py -x -q -k "parse or parser or round_trip or regroup"
# ... and over the parallel inflate (workers chained in file order, consumers parsing the buffer while it fills)
TSAN_OPTIONS=halt_on_error=1 python -m pytest tests/test_pargz.py tests/test_parbz2.py -x -q
