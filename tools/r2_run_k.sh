for pb in 100 50 20 0; do for pe in 100 60; do
  VBH_PRUNE_BACK=$pb VBH_PRUNE_EVERY=$pe timeout -k 10 120 python bench.py --probe --batch 256 2>&1 | grep -E "all utterances|decode of" | sed "s/^/back=$pb every=$pe: /" | cut -c1-260
done; done
