"""MI355X-native hot path of neuroclear (see README.md).  One process-wide setting is made here, before anything can have initialised HIP:
the runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training steps run four to six discriminator
streams beside the main one; with MORE than four hardware queues the persistent 256-workgroup convolution launches of the main stream share
the chip with more side-stream kernels at a time than they can give way to -- 24.0 -> 35.6 ms per Apollo step, 51.3 -> 66 ms per Athena step
(profiles/r06_ab_hw_queues.txt).  An explicit setting in the environment is respected."""
import os

os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
