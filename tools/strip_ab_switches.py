#!/usr/bin/env python3
"""Resolve the A/B and ablation switches of rtl-ws_amd/csrc at their defaults (round 6, one-off).

The product sources used to carry measurement hooks (-DRTLWS_F64_ABL_*, -DRTLWS_X_STAMP, -DRTLWS_STAMP, ...) and
overridable tunables (#ifndef X / #define X default / #endif).  What ships should be what is tested: this script
wrote the product files with every such conditional resolved at its default; the hooks live on as
tools/variants/*.patch (diff product -> instrumented), which `make variant` / `make xvariant` apply to a scratch
copy before compiling with -D flags.  Usage: strip_ab_switches.py <in> <out>
"""
import re
import sys

FLAGS = {  # never defined in the product
    "RTLWS_X_STAMP", "RTLWS_F64_ABL_NOLOAD", "RTLWS_F64_ABL_NOPASS0", "RTLWS_F64_ABL_NOSWAP", "RTLWS_F64_ABL_NOCVT",
    "RTLWS_F64_ABL_NOPASSA", "RTLWS_F64_ABL_NOTWB", "RTLWS_F64_ABL_NOLDS", "RTLWS_F64_ABL_NOPASSB",
    "RTLWS_F64_ABL_NOPOW", "RTLWS_F64_ABL_NOSTORE", "RTLWS_F64_PLAIN_STORE", "RTLWS_ABL_NOMEM", "RTLWS_NO_NT",
    "RTLWS_STAMP", "RTLWS_ABL_NOFFT", "RTLWS_ABL_NOLDS",
}
VALUES = {  # tunables at their defaults
    "RTLWS_X_LDS_ORDER": 0, "RTLWS_V2_PREFETCH": 1, "RTLWS_V2_WINREGS": 0, "RTLWS_V2_NT_STORE": 1,
    "RTLWS_GLDS_AUX": 2, "RTLWS_CIC_TAIL_X4": 0, "RTLWS_FFT16_FMA": 1, "RTLWS_CIC_CT_ROUND": 8,
    "RTLWS_WAVES_BIG": 3, "RTLWS_PREFETCH_4096WIN": 0, "RTLWS_DMA_PF": 1, "RTLWS_V2_DEFAULT": 1,
}
KNOWN = FLAGS | set(VALUES)


def evaluate(expr):
    """True / False when the expression only involves known switches, else None."""
    e = re.sub(r"//.*", "", expr).strip()
    names = set(re.findall(r"[A-Za-z_]\w*", e)) - {"defined"}
    if not names or not names <= KNOWN:
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "0" if m.group(1) in FLAGS else "1", e)
    e = re.sub(r"[A-Za-z_]\w*", lambda m: str(VALUES.get(m.group(0), 0)), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    return bool(eval(e))


def strip(lines):
    out = []
    # stack entries: None = unknown conditional (kept verbatim), else dict(taken=bool any branch taken, live=bool)
    stack = []

    def live():
        return all(s is None or s["live"] for s in stack)

    for ln in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", ln)
        if not m:
            if live():
                out.append(ln)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("ifdef", "ifndef"):
            name = rest.split()[0]
            if name in KNOWN:
                defined = False          # flags: never; tunables: this is their default guard
                v = (not defined) if kind == "ifndef" else defined
                stack.append({"taken": v, "live": v})
            else:
                if live():
                    out.append(ln)
                stack.append(None)
        elif kind == "if":
            v = evaluate(rest)
            if v is None:
                if live():
                    out.append(ln)
                stack.append(None)
            else:
                stack.append({"taken": v, "live": v})
        elif kind == "elif":
            top = stack[-1]
            if top is None:
                if live():
                    out.append(ln)
            else:
                v = evaluate(rest)
                assert v is not None, "mixed chain: " + ln
                top["live"] = (not top["taken"]) and v
                top["taken"] = top["taken"] or v
        elif kind == "else":
            top = stack[-1]
            if top is None:
                if live():
                    out.append(ln)
            else:
                top["live"] = not top["taken"]
                top["taken"] = True
        else:  # endif
            top = stack.pop()
            if top is None and live():
                out.append(ln)
    assert not stack
    return out


if __name__ == "__main__":
    src = open(sys.argv[1]).read().split("\n")
    open(sys.argv[2], "w").write("\n".join(strip(src)))
