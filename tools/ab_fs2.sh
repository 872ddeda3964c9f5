#!/bin/bash
# same-box A/B of the learner kernel forms: CRUX_FS2=0 (k_train_fs) against the default (k_train_fs2), C2 and the C5 shard; prints env-steps/s and us per actor step
for v in 0 1 0 1; do
  echo "CRUX_FS2=$v"; CRUX_FS2=$v bash tools/headline_quick.sh
done
