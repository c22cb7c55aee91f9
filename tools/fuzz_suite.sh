#!/bin/bash
# development: the -m gpu parity suite on OTHER data than the committed seeds' -- DAB_FUZZ_OFFSET shifts every numpy seed the tests use
# (tests/conftest.py).    gpurun -- 'bash tools/fuzz_suite.sh 1 2 3'
for k in "$@"; do
  echo "== offset $k"
  DAB_FUZZ_OFFSET=$k timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py 2>&1 | tail -6
done
