"""The evidence under profiles/ must be machine readable: every kernel table (rocprofv3 `*_kernel_stats*.csv` and the CSV
tables inside the `*_alone.txt` summaries) parses with the csv module at a constant column count, with numeric duration
columns -- round 4 committed a table that `cut -d,` had split inside the kernel names."""
import csv
import glob
import io
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables():
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kernel_stats*.csv"))):
        yield p, open(p).read()
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_alone.txt"))):
        lines = open(p).read().split("\n")
        rows = [l for l in lines if l.startswith('"')]          # the CSV part: quoted kernel names
        if rows:
            yield p, "\n".join(rows)


def test_every_kernel_table_parses_with_a_constant_column_count():
    seen = 0
    for path, text in _tables():
        rows = list(csv.reader(io.StringIO(text)))
        assert rows, path
        head = rows[0]
        assert "Name" in head and "Calls" in head, (path, head)
        i_calls, i_avg = head.index("Calls"), head.index("AverageNs")
        for r in rows[1:]:
            assert len(r) == len(head), (os.path.basename(path), r)
            assert int(r[i_calls]) > 0 and float(r[i_avg]) > 0, (os.path.basename(path), r)
        seen += 1
    assert seen >= 10


def test_counter_files_name_their_own_command():
    import json
    for p in glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_images.json")):
        d = json.load(open(p))
        assert "image" in d["workload"] and "--no-e2e" in d["command"] and "--steps 12" in d["command"] and "image_profile" in d["command"], p
