"""Child process of tests/test_host_logic.py::test_spawned_rank_failure_stops_the_launch: rank 1 dies at once, rank 0 would wait
for ever (as a rank blocked in a barrier does)."""
import os
import sys
import time

if int(os.environ["RANK"]) == 1:
    sys.exit(3)
time.sleep(600)
