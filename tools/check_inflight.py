"""ISA check for kernels that load registers with inline assembly (the compiler does not know those registers are in flight):
between a `global_load_dwordx4 vX` and the next `s_waitcnt vmcnt` no instruction may name a register that is being loaded.
usage: check_inflight.py file.s [symbol-substring ...]; exit code 1 on a hazard."""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(text, want):
    bad = []
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        if want and not any(w in name for w in want):
            continue
        inflight, windows = set(), 0
        for k, line in enumerate(body):
            ls = line.strip()
            if ls.startswith("global_load_dwordx4"):
                inflight |= regs(ls.split()[1].rstrip(","))
                continue
            if not inflight:
                continue
            if ls.startswith("s_waitcnt") and "vmcnt" in ls:
                inflight, windows = set(), windows + 1
                continue
            if re.match(r"^\.LBB\d+_\d+:", ls):
                bad.append((name, k, "block boundary inside a load window"))
            if ls and not ls.startswith((";", ".")):
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", ls):
                    used |= regs(tk)
                if used & inflight:
                    bad.append((name, k, ls[:100]))
        print(f"{name}: {windows} load windows")
    return bad


if __name__ == "__main__":
    hazards = check(open(sys.argv[1]).read(), sys.argv[2:])
    for h in hazards[:20]:
        print("HAZARD", h)
    sys.exit(1 if hazards else 0)
