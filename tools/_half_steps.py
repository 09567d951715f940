"""How many half-steps the one-launch sampler kernel covered in a profiled run of bench.py: a number, or the log of that run
(its last line is the bench line; `sampler_runs.steps_of_every_run`, runs of two steps and more)."""
import json


def half_steps(arg):
    arg = str(arg)
    if arg.isdigit():
        return int(arg)
    line = [l for l in open(arg).read().splitlines() if l.startswith("{")][-1]
    runs = json.loads(line).get("sampler_runs")
    if not runs:
        return 0
    return 2 * sum(st for st in runs["steps_of_every_run"] if st >= 2)
