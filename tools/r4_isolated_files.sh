#!/bin/bash
# Every GPU test file in a process of its own, one after the other: order dependences between tests (a table some earlier test's
# engine happened to fill, round 4's shim bug) do not show in the whole-suite run.
O=gpurun_out; mkdir -p $O; : > $O/isolated_files.txt
for f in tests/test_*.py; do
  r=$(timeout -k 10 600 python -m pytest $f -m gpu -q -p no:cacheprovider 2>&1 | tail -1)
  echo "$f: $r" | tee -a $O/isolated_files.txt
done
