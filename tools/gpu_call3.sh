#!/bin/bash
O=gpurun_out
rm -f $O/parity_observed.json
timeout 2000 python -m pytest tests -m gpu -q -s > $O/b3_tests.log 2>&1
echo "tests rc=$?" >> $O/b3_tests.log
tail -8 $O/b3_tests.log | cut -c1-300; grep -E "PARITY whole_step|sharded vs" $O/b3_tests.log | cut -c1-900
timeout 600 python tools/ab_groups.py > $O/b3_groups.log 2>&1; cat $O/b3_groups.log | cut -c1-300
