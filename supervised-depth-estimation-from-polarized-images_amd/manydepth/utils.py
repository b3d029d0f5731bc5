"""Small helpers used by the trainer (reference manydepth/utils.py)."""


def readlines(filename):
    with open(filename, "r") as f:
        return f.read().splitlines()


def normalize_image(x):
    ma, mi = float(x.max().cpu().data), float(x.min().cpu().data)
    d = ma - mi if ma != mi else 1e5
    return (x - mi) / d


def sec_to_hm(t):
    t = int(t)
    return t // 3600, (t // 60) % 60, t % 60


def sec_to_hm_str(t):
    return "{:02d}h{:02d}m{:02d}s".format(*sec_to_hm(t))
