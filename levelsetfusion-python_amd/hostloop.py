"""Host-side conditions for loops of optimize() calls (the reference's pair loop,
run_hierarchical_optimizer3d_multipair.py:403-432, and bench.py's timed steps).

A 256^3 KillingFusion call takes 1.7 ms; a FULL pass of Python's cyclic collector over the ~10^6 objects torch creates at
import takes ~40 ms, and the allocation counters of an optimize() call trigger one every ~25 calls -- during which the
card idles and drops its clocks (tools/step_times.py).  `parked_collector()` moves everything that
exists when the loop starts into the collector's permanent generation, so that the passes that still run inside the loop
only walk the few objects the calls themselves create: the collector stays ON (cyclic garbage of the loop body is still
found), it just stops re-walking torch.  On exit the parked objects return to the oldest generation."""
import contextlib
import gc


@contextlib.contextmanager
def parked_collector():
    nested = gc.get_freeze_count() > 0  # an enclosing loop has parked already: leave its objects where they are
    if not nested:
        gc.collect()
        gc.freeze()
    try:
        yield
    finally:
        if not nested:
            gc.unfreeze()
