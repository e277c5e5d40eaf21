"""bench.py replays `roofline.traffic` / `roofline.valu` from the committed rocprofv3 counter summary only while the kernel it
measures live still takes what it took in that collection (bench.pmc_fresh); no GPU needed."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_counters_are_replayed_only_for_the_kernel_they_were_collected_on():
    import bench
    stats = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", bench.STATS_FILE)))}
    then_ms = float(stats["k_royale_bloom_h_quad"]["avg_ns"]) * 1e-6
    assert bench.pmc_fresh("royale-bloom-h", then_ms)
    assert bench.pmc_fresh("royale-bloom-h", then_ms * 1.1)
    assert not bench.pmc_fresh("royale-bloom-h", then_ms * 1.3)      # a different kernel (or launch shape) is running
    assert not bench.pmc_fresh("royale-bloom-h", then_ms * 0.6)
    assert not bench.pmc_fresh("no-such-kernel", then_ms)
    # pass 1's launch duration covers its fix-up kernel too
    p1 = (float(stats["k_royale_scan_v_tab"]["avg_ns"]) + float(stats["k_royale_scan_v_fix"]["avg_ns"])) * 1e-6
    assert bench.pmc_fresh("royale-scanlines-v", p1)
    # the traffic figure itself: 2 x FETCH_SIZE + WRITE_SIZE of the quad kernel, within 10 % of pass 10's algorithmic 4F + halation
    t = bench.pmc_traffic("royale-bloom-h", bench.PMC_FRAMES_PER_LAUNCH)
    assert t is not None and 0.9 < t / (128 * (4 * 1920 * 1080 * 4 + 320 * 240 * 4)) < 1.1
    assert bench.pmc_traffic("royale-bloom-h", 64.0) is None           # another launch shape: withheld
