"""Statistics of an interleaved N-way bench.py comparison (tools/abn_bench.sh writes gpurun_out/<tag>_<arm>_<round>.json).

    python tools/ab_stats.py <tag> [dir]

Per arm: mean +- sd of `value` over the rounds.  Per arm k > 0 against arm 0: the PAIRED differences d_i = v_k,i - v_0,i
of the interleaved rounds (a box's drift over the minutes of one call is common to both arms of a round), their mean and
the standard error sd(d) / sqrt(n).  The keep rule: a difference is called only at
|mean d| >= 2 standard errors AND n >= 5; anything else prints "no verdict".
"""
import glob
import json
import math
import os
import re
import sys


def load(tag, d):
    arms = {}
    for f in glob.glob(os.path.join(d, f"{tag}_*_*.json")):
        m = re.match(rf"{re.escape(tag)}_(\d+)_(\d+)\.json$", os.path.basename(f))
        if not m:
            continue
        try:
            line = open(f).read().strip().splitlines()[-1]
            rec = json.loads(line)
        except (IndexError, ValueError):
            continue
        arms.setdefault(int(m.group(1)), {})[int(m.group(2))] = rec
    return arms


def mean_sd(v):
    n = len(v)
    mu = sum(v) / n
    sd = math.sqrt(sum((x - mu) ** 2 for x in v) / (n - 1)) if n > 1 else float("nan")
    return mu, sd


def main():
    tag = sys.argv[1]
    d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
    arms = load(tag, d)
    if not arms:
        print(f"no {tag}_<arm>_<round>.json under {d}")
        return 1
    labels = {}
    lf = os.path.join(d, f"{tag}_arms.txt")
    if os.path.exists(lf):
        for i, line in enumerate(open(lf).read().splitlines()):
            labels[i] = line
    print(f"# {tag}: interleaved rounds, `value` (point-clouds/s) of bench.py; keep rule: |paired mean difference| >= 2 standard errors, n >= 5")
    print("| arm | env | n | mean | sd | min | max | vs arm 0: paired mean d +- se | d / se | verdict |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    base = arms.get(0, {})
    for k in sorted(arms):
        vals = [arms[k][i]["value"] for i in sorted(arms[k])]
        mu, sd = mean_sd(vals)
        cmp_, ratio, verdict = "", "", "baseline" if k == 0 else ""
        if k != 0 and base:
            common = sorted(set(arms[k]) & set(base))
            ds = [arms[k][i]["value"] - base[i]["value"] for i in common]
            if len(ds) >= 2:
                dm, dsd = mean_sd(ds)
                se = dsd / math.sqrt(len(ds))
                cmp_ = f"{dm:+.2f} +- {se:.2f} ({100 * dm / mean_sd([base[i]['value'] for i in common])[0]:+.2f} %)"
                z = abs(dm) / se if se > 0 else float("inf")
                ratio = f"{z:.1f}"
                if len(ds) < 5:
                    verdict = "no verdict (n < 5)"
                elif z < 2.0:
                    verdict = "no verdict (< 2 se)"
                else:
                    verdict = "FASTER" if dm > 0 else "SLOWER"
        print(f"| {k} | {labels.get(k, '')} | {len(vals)} | {mu:.1f} | {sd:.1f} | {min(vals):.1f} | {max(vals):.1f} | {cmp_} | {ratio} | {verdict} |")
    return 0


if __name__ == "__main__":
    sys.exit(main())
