"""MI355X-native page-inference hot path of RapidDoc: HIP kernels + C-ABI library under `csrc/`, the host-side mirror
of the reference's interfaces above it."""
import os

# One page batch keeps about fourteen HIP streams busy (eight recogniser streams, the neck + CTC tail, det, layout, the uploader's
# copy stream, one per pool worker).  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues - four by default -
# and work of streams that share a queue is dispatched in order: the next batch's layout backbone ended up in front of this batch's
# DB post-process and the host waited 15 ms for a result that was ready ("front prefetch").  Sixteen queues give every
# stream its own.  Only effective when set before the HIP runtime initialises, i.e. import this package (or set the variable) before
# the first torch.cuda call; an explicit setting in the environment wins.
HW_QUEUES_SET_BY_IMPORT = "GPU_MAX_HW_QUEUES" not in os.environ      # True: this import changed the process environment
if HW_QUEUES_SET_BY_IMPORT:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
    # a side effect on the embedding process: say so, once (logging: silent unless the application configured a handler at INFO)
    import logging
    logging.getLogger("rapiddoc_amd").info("rapiddoc_amd: GPU_MAX_HW_QUEUES was unset - set to 16 for this process (one hardware queue per "
                                           "HIP stream of the page pipeline; export the variable yourself to choose another value)")
