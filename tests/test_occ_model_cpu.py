"""tests/experiments/occ_model.c: the CPU model of an exact-order PARALLEL insert (plan a window of inserts
against one snapshot, commit in id order, validate each plan against the journal of row changes since its
snapshot).  The model exits 0 only if its graph is row-for-row the plain oracle's, i.e. if the validation
rules are sound; DESIGN.md section 4.2c quotes its yield figures.  Kept under test so the analysis stays
reproducible."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_occ_model_reproduces_the_serial_graph(tmp_path):
    exe = str(tmp_path / "occ_model")
    src = os.path.join(ROOT, "tests", "experiments", "occ_model.c")
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-w", "-o", exe, src, "-lm", "-lpthread"])
    for args in (["1500", "160", "16", "32", "6", "40"], ["800", "128", "32", "16", "4", "24"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "graphs IDENTICAL" in r.stdout
        m = re.search(r"commits/round=([0-9.]+)", r.stdout)
        assert m and float(m.group(1)) >= 1.0


def test_parallel_validated_commits_reproduce_the_serial_graph(tmp_path):
    """PAR=1: the commits of a round are applied in validated GROUPS instead of one by one -- every
    window node runs its whole commit against the graph as it stands (a dry run in a private overlay of the rows it
    rewrites), and node j joins the group of the nodes before it iff none of their deltas is relevant to anything j
    read (link plan, the speculative records it used, the select_neighbors it recomputed) and none is on a row j
    changed.  The model's graph must stay the serial oracle's; without the changed-row rule (PAR_NOROWCHECK, unsound on
    purpose) it must not."""
    exe = str(tmp_path / "occ_model")
    src = os.path.join(ROOT, "tests", "experiments", "occ_model.c")
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-w", "-o", exe, src, "-lm", "-lpthread"])
    env = dict(os.environ, PAR="1")
    for args in (["1500", "160", "16", "32", "6", "40"], ["800", "128", "32", "16", "4", "24"], ["3000", "400", "32", "16", "8", "32"],
                 ["2000", "300", "24", "8", "5", "24"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "graphs IDENTICAL" in r.stdout
        m = re.search(r"([0-9.]+) nodes per group", r.stdout)
        assert m and float(m.group(1)) > 1.0               # groups do form
    r = subprocess.run([exe, "2000", "300", "24", "8", "5", "24"], capture_output=True, text=True, timeout=600,
                       env=dict(env, PAR_NOROWCHECK="1"))
    assert r.returncode != 0 and "graphs IDENTICAL" not in r.stdout


def test_look_ahead_plans_made_during_the_commits_are_sound_under_the_hot_row_rule(tmp_path):
    """AHEAD (round 6, DESIGN.md 4.2g / docs/history.md): the nodes that enter the window next round are planned WHILE this
    round's commits are applied -- modelled by planning them against the graph as it stands after the round's first group --
    with the journal position at the START of the commits as their snapshot, and every delta journalled during those
    commits that sits on a row such a plan read voids the read outright (the row may have been seen before, after, or
    torn).  The graphs must stay the serial oracle's; the model also says what the rule costs (most look-ahead plans are
    voided on small indexes), which is why the overlap was not built for the GPU."""
    exe = str(tmp_path / "occ_model")
    src = os.path.join(ROOT, "tests", "experiments", "occ_model.c")
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-w", "-o", exe, src, "-lm", "-lpthread"])
    env = dict(os.environ, PAR="1", AHEAD="1.3")
    for args in (["1500", "200", "16", "32", "6", "40"], ["3000", "400", "24", "16", "8", "32"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "graphs IDENTICAL" in r.stdout
        m = re.search(r"look-ahead plans \(made during the commits\): ([0-9]+), of which voided by a hot delta on a row they read: link plan ([0-9]+)", r.stdout)
        assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0


def test_select_after_search_is_the_head_of_W(tmp_path):
    """tests/experiments/select_head.c: the reference's insert() with the full select_neighbors never returns
    anything but the m nearest of W (what the engine's plan kernels use instead of the extension)."""
    exe = str(tmp_path / "select_head")
    src = os.path.join(ROOT, "tests", "experiments", "select_head.c")
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-w", "-o", exe, src, "-lm", "-lpthread"])
    for args in (["4000", "32", "16", "200"], ["3000", "16", "5", "16"], ["2500", "8", "4", "8"], ["2500", "8", "5", "16", "dup"],
                 ["2000", "16", "8", "8"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "differing from the m nearest of W: 0" in r.stdout


def test_speculated_delete_model_reproduces_the_serial_delete(tmp_path):
    """tests/experiments/del_model.c: HNSW.NODE.DEL with every neighbour's re-selection computed on the graph as
    it stands before the delete, then validated against the delete's own journal and applied in the reference's
    order (what k_occ_del_list / k_occ_shrinks / k_occ_del_commit do on the GPU) gives the serial delete's graph --
    and does NOT once the rule "a selection that did not fill its list counts every change as relevant" is dropped
    (m = 2), which is the rule the GPU campaign found missing."""
    exe = str(tmp_path / "del_model")
    src = os.path.join(ROOT, "tests", "experiments", "del_model.c")
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-w", "-o", exe, src, "-lm", "-lpthread"])
    for args in (["1500", "300", "32", "8", "40"], ["800", "400", "16", "2", "8"], ["600", "300", "128", "16", "64"],
                 ["400", "250", "8", "24", "48"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "graphs IDENTICAL" in r.stdout
        m = re.search(r"speculative result used ([0-9.]+)", r.stdout)
        assert m and float(m.group(1)) > 0.0
    r = subprocess.run([exe, "800", "400", "16", "2", "8", "norule"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "graphs DIFFER" in r.stdout
