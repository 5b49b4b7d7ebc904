"""Writes tests/golden/checkpoint_policy.json by driving the REFERENCE's CheckpointManager
(frl/training/representation/checkpointing.py:22-150; torch-agnostic, importable in the build container) over metric sequences and
recording the directory listing after every epoch.  Only the listings travel.

    python tests/golden/make_checkpoint_golden.py
"""
import json
import logging
import os
import sys
import tempfile
from types import SimpleNamespace

sys.path.insert(0, "/root/reference/frl")
from training.representation.checkpointing import CheckpointManager  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NAN = float("nan")
CASES = [
    {"name": "min_top3", "cfg": dict(monitor="val/loss", mode="min", save_last=True, save_every_n_epochs=2, save_top_k=3, monitor_start_epoch=0),
     "values": [5.0, 4.0, 4.5, NAN, 3.0, 3.0, 6.0, 2.5, 2.75, 1.0]},
    {"name": "max_top2_late_start", "cfg": dict(monitor="val/score", mode="max", save_last=False, save_every_n_epochs=3, save_top_k=2, monitor_start_epoch=2),
     "values": [9.0, 8.0, 0.1, 0.5, 0.4, NAN, 0.5, 0.7, 0.2]},
    {"name": "min_top1", "cfg": dict(monitor="m", mode="min", save_last=True, save_every_n_epochs=100, save_top_k=1, monitor_start_epoch=1),
     "values": [0.5, 1.0, 0.9, 0.95, 0.1, 0.1]},
]


def main():
    log = logging.getLogger("golden")
    out = []
    for case in CASES:
        with tempfile.TemporaryDirectory() as d:
            def save_fn(state, path):
                with open(path, "w") as fh:
                    json.dump({k: (None if isinstance(v, float) and v != v else v) for k, v in state.items()}, fh)

            def load_fn(path):
                with open(path) as fh:
                    s = json.load(fh)
                return {k: (NAN if v is None else v) for k, v in s.items()}
            mgr = CheckpointManager(d, SimpleNamespace(**case["cfg"]), log, save_fn, load_fn)
            listings = []
            for ep, v in enumerate(case["values"]):
                metrics = {case["cfg"]["monitor"]: v}
                mgr.save(ep, {"epoch": ep + 1, **metrics}, metrics)
                listings.append(sorted(os.listdir(d)))
            # auto-resume: a fresh manager rebuilds its top-k list from disk and continues
            mgr2 = CheckpointManager(d, SimpleNamespace(**case["cfg"]), log, save_fn, load_fn)
            mgr2.restore_top_k()
            extra = [case["values"][-1] * 0.5 if case["cfg"]["mode"] == "min" else case["values"][-1] + 5.0]
            for j, v in enumerate(extra):
                ep = len(case["values"]) + j
                metrics = {case["cfg"]["monitor"]: v}
                mgr2.save(ep, {"epoch": ep + 1, **metrics}, metrics)
                listings.append(sorted(os.listdir(d)))
            out.append({"name": case["name"], "cfg": case["cfg"], "values": [None if v != v else v for v in case["values"]],
                        "resume_values": extra, "listings": listings})
    with open(os.path.join(HERE, "checkpoint_policy.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
