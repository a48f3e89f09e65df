"""bench_c1.py's job (BASELINE config 1 / 4) as a test, at a small size: captures generated on the device, decoded from their first
sample to TS by worker threads (cu8 fused into the receiver, packed decisions, deconvol_sync on packed symbols, mpeg_sync,
deinterleaver, rs_decoder, derandomizer), every capture's TS checked against the reference binary's TS for the same IQ (or, where
oracle/_ref is not built, against the transmitted sequence)."""
import sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile,warm", [(2048, 512), (1024, 512)])
def test_c1_job_decodes_every_capture_to_the_reference_ts(capi, tile, warm):
    sys.path.insert(0, ROOT)
    import bench_c1
    job = bench_c1.ChainJob(capi, 0, 3, 6, 2, tile, warm, seed0=4000)        # (the round-6 engine: tests/test_gpu_capture_batch.py)
    try:
        job.run(1)
        consumed = job.run(2, timed=True)
        assert consumed == 2 * 3 * (job.n - 1)            # every decode consumed the whole capture (128-sample chunks, 1 of read-ahead)
        v = job.verify()
        assert v["pass"], v
        assert all(c["ts_packets"] > 3000 and c["same_count_every_step"] for c in v["per_capture"])
        ms, n = job.tile_kernel_ms()
        assert n == 6 and ms > 0
    finally:
        job.close()
