import os

import torch

__all__ = ['auto_device', 'pin_host_threads', 'launch_affinity']

# the CPU set this process was started with (taskset, cgroup cpuset, the launcher): recorded at import, because the HIP
# runtime may widen the mask to every CPU of the box when it initialises (seen on the MI355X boxes: `taskset -c 0 python …`
# reads 0-255 after torch.cuda.init(), on some boxes and not on others — profiles/r05_experiments/host_pin.txt)
_LAUNCH_AFFINITY = frozenset(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else frozenset()


def auto_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def launch_affinity():
    return _LAUNCH_AFFINITY


def pin_host_threads(local_rank=0, cores=None):
    """Keep this rank's enqueuing threads on a few CPUs of the set the process was launched with.

    One training step is enqueued by ONE busy thread at a time (the Python thread forward, autograd's device thread
    backward); on a box that shows 256 logical CPUs under a 16-core quota they migrate, and the step costs the host 3-4 ms
    more than on two CPUs.  `cores` CPUs per rank (default: EVK_HOST_CORES, else 0 = only undo a widening of
    the launch mask by the runtime), rank r taking the r-th group of the launch set — if the set holds a group for every local rank
    (LOCAL_WORLD_SIZE), otherwise nobody is narrowed; threads created afterwards (autograd's,
    the side stream's callbacks) inherit the mask.  Returns the CPU set now in force.  No-op where the OS has no affinity
    call."""
    if not hasattr(os, 'sched_setaffinity') or not _LAUNCH_AFFINITY:
        return frozenset()
    if cores is None:
        cores = int(os.environ.get('EVK_HOST_CORES', '0') or 0)
    allowed = sorted(_LAUNCH_AFFINITY)
    # every local rank gets its OWN group, or nobody is narrowed (two ranks on the same four CPUs would be worse than none pinned)
    ranks = max(int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1), local_rank + 1)
    if cores > 0 and len(allowed) > cores and len(allowed) >= cores * ranks:
        want = allowed[local_rank * cores:(local_rank + 1) * cores]
    else:
        want = allowed
    try:
        if frozenset(os.sched_getaffinity(0)) != frozenset(want):
            os.sched_setaffinity(0, want)
    except OSError:
        pass
    return frozenset(os.sched_getaffinity(0))
