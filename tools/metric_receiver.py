#!/usr/bin/env python3
"""Forward a metrics JSON-lines file (written by RLGPC::MetricSender / rlgymppo_cpp_amd.learner.MetricSender) to wandb.

The reference embeds an interpreter and calls python_scripts/metric_receiver.py: init(project, group, name, id) ->
wandb.init(..., resume="allow"), add_metrics(dict) -> run.log(dict).  Here the learner only appends lines to
metrics/<project>/<run id>.jsonl; this side-car makes the same two wandb calls while following the file, so training never
depends on Python or on the network.

usage: tools/metric_receiver.py metrics/<project>/<run id>.jsonl [--once] [--dry-run]
  --once     forward what is in the file and exit (default: keep following it like `tail -f`)
  --dry-run  print what would be logged instead of importing wandb
"""
import argparse
import json
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--once", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--record-calls", action="store_true", help="(tests) make the calls on a stand-in for wandb and print them as one JSON line")
    args = ap.parse_args()
    run = None
    wandb = None
    recorded = []
    if args.record_calls:
        class _Run:
            def log(self, d): recorded.append(["log", {k: (repr(v) if isinstance(v, float) else v) for k, v in d.items()}])

        class _Wandb:
            @staticmethod
            def init(**kw): recorded.append(["init", dict(kw)]); return _Run()
        wandb = _Wandb
    elif not args.dry_run:
        try:
            import wandb
        except Exception as e:  # same failure mode as the reference receiver: say which interpreter lacks wandb
            raise SystemExit(f"FAILED to import wandb with {sys.executable}: {e!r} (use --dry-run to only print)")
    with open(args.path) as f:
        buf = ""
        while True:
            chunk = f.readline()
            if not chunk:
                if args.once:
                    break
                time.sleep(0.5)
                continue
            buf += chunk
            if not buf.endswith("\n"):
                continue   # a line still being written
            line, buf = buf.strip(), ""
            if not line:
                continue
            rec = json.loads(line)
            if "_run" in rec:
                d = rec["_run"]
                if args.dry_run:
                    print("init", d)
                else:
                    run = wandb.init(project=d["project"], group=d["group"], name=d["name"], id=d["id"], resume="allow")
                continue
            # a NaN metric is a null in the file (JSON has no NaN); the reference hands wandb the NaN itself (MetricSender.cpp:31-44)
            rec = {k: (float("nan") if v is None else (float(v) if isinstance(v, int) and not isinstance(v, bool) else v)) for k, v in rec.items()}
            if args.dry_run:
                print("log", rec)
            elif run is not None:
                run.log(rec)
    if args.record_calls:
        print(json.dumps(recorded))


if __name__ == "__main__":
    main()
