"""MI355X-native UNOPose forward hot path."""
import os

# The runner keeps up to ten HIP streams alive (two pipeline streams, a features + matching pair, a helper stream each, the default stream
# and its helper), and the HIP runtime folds streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) for the life of the process:
# two streams that share a queue serialise.  Measured in round 6 (same box, bench.py's bounded legs): training step 120.8 -> 111.4 ms,
# 224 x 224 contract 2039 -> 2120 pairs/s (bf16) and 896 -> 947 (fp32) with 8 queues; once the fp32 path was pipelined as well the tenth
# stream wrapped onto the default stream's queue again (training leg back to 119.6 ms) and 16 queues cured it (109.8).  The headline step is unchanged.  A default only:
# a value the user has set wins, and it takes effect only if the package is imported before the process first touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
