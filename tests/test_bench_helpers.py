"""Host-side pieces of bench.py that need no GPU: the exchange model of the multi-GPU protocols (sizes taken from
columbiaimagesearch_amd/distributed.py), the provenance record of replayed PMC summaries, the binding-resource roofline."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_exchange_model_sizes_follow_the_protocols():
    import bench
    from columbiaimagesearch_amd.distributed import exchange_stride, route_capacity
    m = bench.exchange_model(8, 8192, 1024, 100, 1024)
    ag, rt = m["allgather"], m["routed"]
    assert ag["payload_bytes_per_rank"] == exchange_stride(8192, 100, 8) * 32 and ag["counts_bytes_per_rank"] == 8192 * 4
    assert ag["received_bytes_per_rank"] == 7 * (ag["payload_bytes_per_rank"] + ag["counts_bytes_per_rank"])
    assert rt["query_block_bytes_per_peer"] == route_capacity(1024, 256, 8) * 1024
    assert ag["exchange_bytes_per_step"] > 0 and rt["exchange_bytes_per_step"] > rt["query_bytes_sent_per_rank"]
    # per-peer share / link rate + fixed latencies: tens to hundreds of microseconds at these sizes, and the weak form moves 8 x the rows
    assert 50.0 < ag["projected_us"] < 250.0 and 100.0 < rt["projected_us"] < 250.0
    weak = bench.exchange_model(8, 8 * 8192, 8192, 100, 1024)["routed"]
    assert weak["query_block_bytes_per_peer"] == 8 * rt["query_block_bytes_per_peer"] and weak["projected_us"] > rt["projected_us"]
    one = bench.exchange_model(1, 8192, 8192, 100, 1024)
    assert one["allgather"]["received_bytes_per_rank"] == 0


def test_pmc_source_names_a_stale_summary(tmp_path):
    import bench
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from pmc_stamp import kernel_sources_sha1
    have = kernel_sources_sha1()
    assert set(have) >= {"lopq_scan3.hip", "lopq_search.hip", "lopq_stream.hip"} and all(v and len(v) == 40 for v in have.values())
    p = os.path.join(REPO, "profiles", "scan_binding_c4.json")
    fresh = {"kernel": "void k_adc_scan4<8, 2, 4, 4, 1016>", "commit": "abc", "kernel_sources_sha1": dict(have)}
    s = bench.pmc_source(p, fresh)
    assert s == {"file": "profiles/scan_binding_c4.json", "commit": "abc", "kernel": fresh["kernel"], "sources_match_this_build": True}
    stale = dict(fresh, kernel_sources_sha1=dict(have, **{"lopq_scan3.hip": "0" * 40}))
    s = bench.pmc_source(p, stale)
    assert s["sources_match_this_build"] is False and s["changed_since"] == ["lopq_scan3.hip"]
    assert bench.pmc_source(p, {"kernel": "k"})["sources_match_this_build"] is None   # a summary from before round 6


def test_committed_pmc_summaries_carry_their_provenance():
    for name in ("scan_binding_c4.json", "scan_traffic_c4.json", "scan_traffic_c4x.json"):
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert d.get("commit") and isinstance(d.get("kernel_sources_sha1"), dict), name


def test_binding_frac_is_a_fraction():
    import bench
    # the C4 launch of round 6: 8192 queries x 40834 candidates, 0.2296 ms, 20 launches' worth of totals
    launches = 20
    cand = 8192.0 * 40834 * launches
    b = bench.scan_binding("c4", "k_adc_scan4", cand, 8192.0 * 1.3 * launches, 8, 0.2296e-3 * launches, launches)
    assert 0.0 < b["binding_frac"] <= 1.0 and b["binding_resource"] in ("lds_gather", "valu_issue", "hbm")
    assert set(b["binding_min_ms_per_launch"]) == {"lds_gather", "valu_issue", "hbm"}
    assert all(v <= 0.2296 * 1.0001 for v in b["binding_min_ms_per_launch"].values())
    # a launch faster than its own minimum cannot be reported as more than the whole resource
    assert bench.scan_binding("c4", "k_adc_scan4", cand, 8192.0 * 1.3 * launches, 8, 0.01e-3 * launches, launches)["binding_frac"] == 1.0
