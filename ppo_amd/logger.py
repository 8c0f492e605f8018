"""Metrics logger — host mirror of the reference's rl/logger.py API used by Runner / ppo.train
(`watch`, `watch_mean`, `watch_full`, `watch_stats`, `record_step`, `export_to_csv`,
`print_variables`, `info/warn/important/log`; rl/logger.py:125-346).

Plain Python, no device work.  The one behavioural difference from the reference is deliberate:
values arrive as host floats that the Runner fetched in ONE device->host copy per iteration,
instead of ~12 `float(tensor)` syncs per minibatch (rl/logger.py:58-67).
"""
import collections
import csv
import math
import os
import sys
import time


class LogVariable:
    def __init__(self, name, history_length=1, type="float", display_width=None, display_name=None, display_precision=2):
        self.name = name
        self.display_name = display_name or name
        self.display_width = display_width
        self.display_precision = display_precision
        self.history = collections.deque(maxlen=max(1, history_length))
        self.type = type

    def add(self, value):
        self.history.append(float(value))

    @property
    def value(self):
        if not self.history:
            return None
        if self.type == "stats":
            h = list(self.history)
            mean = sum(h) / len(h)
            return (mean, math.sqrt(sum((x - mean) ** 2 for x in h) / len(h)), min(h), max(h))
        return sum(self.history) / len(self.history)


class Logger:
    def __init__(self, csv_path=None, txt_path=None, quiet=False):
        self.vars = collections.OrderedDict()
        self.rows = []
        self.csv_path = csv_path
        self.txt_path = txt_path
        self.quiet = quiet
        self._t0 = time.time()

    # --- variable registration / updates
    def add_variable(self, v: LogVariable):
        self.vars[v.name] = v

    def _get(self, key, history_length, type_="float", **kw):
        v = self.vars.get(key)
        if v is None:
            v = LogVariable(key, history_length, type_, **{k: kw[k] for k in ("display_width", "display_name", "display_precision") if k in kw})
            self.vars[key] = v
        return v

    def watch(self, key, value, **kw):
        self._get(key, 1, **kw).add(value)

    def watch_mean(self, key, value, history_length=10, **kw):
        self._get(key, history_length, **kw).add(value)

    def watch_full(self, key, value, history_length=100, **kw):
        self._get(key, history_length, "stats", **kw).add(value)

    def watch_stats(self, key, values, **kw):
        import numpy as np
        a = np.asarray(values, dtype=np.float64).ravel()
        if a.size:
            for suffix, val in (("mean", a.mean()), ("std", a.std()), ("min", a.min()), ("max", a.max())):
                self.watch(f"{key}_{suffix}", val, **kw)

    def watch_mean_std(self, key, values, **kw):
        import numpy as np
        a = np.asarray(values, dtype=np.float64).ravel()
        if a.size:
            self.watch(f"{key}_mean", a.mean(), **kw)
            self.watch(f"{key}_std", a.std(), **kw)

    def __getitem__(self, key):
        return self.vars[key].value

    def __contains__(self, key):
        return key in self.vars

    # --- rows
    def record_step(self):
        row = {}
        for k, v in self.vars.items():
            val = v.value
            if isinstance(val, tuple):
                row[k] = val[0]
            elif val is not None:
                row[k] = val
        self.rows.append(row)

    def export_to_csv(self, path=None):
        path = path or self.csv_path
        if not path or not self.rows:
            return
        keys = []
        for r in self.rows:
            for k in r:
                if k not in keys:
                    keys.append(k)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(self.rows)

    def print_variables(self, include_header=True, file=None):
        shown = [(k, v) for k, v in self.vars.items() if not k.startswith("*") and v.display_width != 0 and v.value is not None]
        names = " ".join(f"{v.display_name[:12]:>12}" for _, v in shown)
        vals = " ".join(f"{(v.value[0] if isinstance(v.value, tuple) else v.value):>12.4g}" for _, v in shown)
        if include_header:
            self.log(names, file=file)
        self.log(vals, file=file)

    # --- text
    def log(self, s="", level="info", file=None):
        if not self.quiet:
            print(s, file=file or sys.stdout, flush=True)
        if self.txt_path:
            with open(self.txt_path, "a") as f:
                f.write(str(s) + "\n")

    def info(self, s):
        self.log(s, "info")

    def warn(self, s):
        self.log("[warn] " + str(s), "warn")

    def important(self, s):
        self.log(s, "important")

    def error(self, s):
        self.log("[error] " + str(s), "error")

    def debug(self, s):
        pass
