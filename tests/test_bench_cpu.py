"""bench.py's bookkeeping that needs no GPU: the committed counter summaries it prices the rooflines with (profiles/rNN_pmc_*.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_counter_summaries_are_found_and_template_instances_are_added_up(tmp_path, monkeypatch):
    import bench
    # the newest committed summary of every workload exists and names the kernels the rooflines ask for
    for workload, kernel in (("refine", "k_global"), ("refine", "k_local"), ("reconstruct", "k_insert_bricks"), ("sva", "k_sva_eval"), ("csp", "k_csp_eval")):
        name = bench.pmc_latest(workload)
        assert name and os.path.exists(os.path.join(ROOT, "profiles", name)), workload
        e, meta = bench.pmc_entry(name, kernel, merge=True)
        assert e and meta.get("particles"), (workload, kernel)
        assert e["SQ_INSTS_VALU"]["sum"] > 0 and "FETCH_SIZE" in e and "WRITE_SIZE" in e, (workload, kernel)
    # two instances of one kernel template: their counters are sums, per_dispatch follows
    d = {"_meta": {"particles": 4}, "ppm::k_x<0>": {"C": {"sum": 10.0, "dispatches": 2, "per_dispatch": 5.0}},
         "ppm::k_x<6>": {"C": {"sum": 30.0, "dispatches": 3, "per_dispatch": 10.0}}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    json.dump(d, open(tmp_path / "profiles" / "t.json", "w"))
    e, _ = bench.pmc_entry("t.json", "k_x", merge=True)
    assert e["C"] == {"sum": 40.0, "dispatches": 5, "per_dispatch": 8.0}
    e, _ = bench.pmc_entry("t.json", "k_x<6>")
    assert e["C"]["sum"] == 30.0
