#!/bin/bash
# round 3, session ak: does the speed level of an allocation (probe_alloc_modes.py: 23.1 / 25.0 / 26-27) move with the relative
# alignment of the queue arrays?  PT_X_SKEW = bytes between consecutive arrays' starts inside their allocations
for sk in 0 4352 0 65792 0 1048832 0 4352; do echo "skew $sk:"; PT_X_SKEW=$sk python scripts/probe_alloc_modes.py 5 | cut -c1-230; done
