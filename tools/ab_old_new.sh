#!/bin/bash
# Same-box A/B of two TREES (the committed one unpacked and built under _ab_old/, and the working tree):
#   bash tools/ab_old_new.sh '<command run from the tree root>'   — twice each, alternating
CMD="$1"
for rep in 1 2; do for t in _ab_old .; do
  echo "== $t"; (cd $t && bash -c "$CMD" 2>/dev/null | tail -${2:-3})
done; done
