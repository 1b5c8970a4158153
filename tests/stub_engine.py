"""Stand-in for blaze_amd.engine.Engine in the CPU test of bench.py's launch logic (tests/test_dist_cpu.py):
no GPU, no scan -- it counts rows and pretends every non-blank row was assigned to barcode 0.  Test infrastructure
only; bench.py labels a line produced with it as such."""
import time

import numpy as np


class StubEngine:
    def __init__(self, device=0):
        self.device, self.n_wl, self.rows, self._counts = device, 0, [None, None], None

    def set_option(self, *a):
        pass

    def device_identity(self):
        """bench.py counts the distinct GPUs of a run from these: $STUB_GPU_OF_RANK = comma list, one pretend GPU per rank."""
        import os
        ids = os.environ.get("STUB_GPU_OF_RANK", "")
        rank = int(os.environ.get("RANK", "0"))
        return ids.split(",")[rank] if ids else f"stub-gpu-{rank}"

    def set_whitelist_codes16(self, m):
        self.n_wl = int(m.shape[0])
        self._counts = np.zeros(self.n_wl + 2, dtype=np.int64)

    def stage(self, slot, ext, el, bo, bl):
        self.rows[slot] = np.asarray(bl)

    def sync(self, slot):
        pass

    def run(self, slot, max_ed):
        bl = self.rows[slot]
        time.sleep(0.001)
        a = int((bl > 0).sum())
        self._counts[0] += a
        self._counts[-2] += a
        self._counts[-1] += bl.size

    def counts_reset(self):
        self._counts[:] = 0

    def counts(self):
        return self._counts.copy()

    def timing(self, slot):
        n = int(self.rows[slot].size)
        return dict(total_ms=1.0, classify_ms=0.2, scan_ms=0.6, generic_ms=0.0, hist_ms=0.2, candidates_ms=0.3, verify_ms=0.3,
                    rows=n, rows_bitparallel=n // 3, rows_generic=0, n_whitelist=self.n_wl, rows_redo=0)

    def valu_peak_probe(self, iters=0):
        return 1e12, 1.0

    def submit(self, slot, ext, el, bo, bl, max_ed):
        self.stage(slot, ext, el, bo, bl)

    def wait(self, slot):
        n = self.rows[slot].size
        return np.full(n, -1, np.int32), np.zeros(n, np.int8), np.zeros(n, np.int8), np.zeros(n, np.uint8)
