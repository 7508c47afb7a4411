#!/bin/bash
O=gpurun_out
rm -f $O/parity_observed.json
timeout 2000 python -m pytest tests -m gpu -q --durations=45 > $O/b4_tests.log 2>&1
echo "tests rc=$?" >> $O/b4_tests.log
tail -60 $O/b4_tests.log | cut -c1-200
