"""Kernel names as the profile summaries key them."""
import re


def kernel_key(name):
    """`k_xyz` for every instance of a kernel template -- except the forward renderer, whose keep-state instance (frames that
    keep backward state walk per-TILE lists: other bytes, other duration) is reported as k_render_forward_b_keep."""
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    k = re.sub(r"^void ", "", k).split("(")[0]
    base = k.split("<")[0].split("::")[-1]
    if base == "k_render_forward_b" and "<" in k:
        targs = [a.strip() for a in k.split("<", 1)[1].rsplit(">", 1)[0].split(",")]
        if len(targs) >= 2 and targs[1] == "true":
            base += "_keep"
    return base
