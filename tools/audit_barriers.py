"""Static audit of a gfx950 assembly listing: LDS reads that can still be in flight at an s_barrier.

The rule the kernels follow: every LDS fragment a wave reads is WAITED FOR before the
barrier that hands the fragment's source (a weight slot, a window, a strip) to its next writer.  A
read sunk behind an MFMA block by the scheduler moves its s_waitcnt with it -- the failure of round 5
(one wrong tile in several hundred launches).  This tool finds every place where that CAN happen, from
the compiler's own output, without a GPU:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s centernet_amd/csrc/<file>.hip
    python tools/audit_barriers.py k.s [kernel-name substring]

Per kernel it runs a forward data-flow over the basic blocks.  The state is the queue of outstanding
LGKM operations (ds_read / ds_write / s_load ...) in issue order, capped at the counter's 15;
`s_waitcnt lgkmcnt(n)` keeps the youngest n.  At a join the longer queue wins (conservative for "can a
read be outstanding").  For every s_barrier it prints the ds_reads that may be outstanding, where their
destination registers are first used, and what the next LDS writers behind the barrier are (ds_write,
global_load_lds) -- the reader decides whether the barrier frees that read's source.

Exit status 0 always: it is a review aid, `tests/test_isa_audit.py` holds the committed expectations.
"""
import re
import sys
from collections import defaultdict

LGKM_RE = re.compile(r"^\s*(ds_\w+|s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime|s_sendmsg\w*)\b")
WAIT_RE = re.compile(r"^\s*s_waitcnt\b(.*)")
LABEL_RE = re.compile(r"^(\.LBB\d+_\d+):")
BR_RE = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
END_RE = re.compile(r"^\s*s_endpgm")
REG_RE = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
CAP = 15


def regs_of(text):
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add(m.group(1) + str(i))
        else:
            out.add(m.group(4) + m.group(5))
    return out


def split_kernels(lines):
    """yield (name, first, last) for every function body of the listing"""
    name, start = None, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L") and name is None:
            name, start = m.group(1), i + 1
        elif name and l.strip().startswith(".Lfunc_end"):
            yield name, start, i
            name = None


def analyse(lines, first, last):
    # basic blocks
    heads = {first}
    label_at = {}
    for i in range(first, last):
        m = LABEL_RE.match(lines[i])
        if m:
            label_at[m.group(1)] = i
            heads.add(i)
        if BR_RE.match(lines[i]) or END_RE.match(lines[i]):
            heads.add(i + 1)
    heads = sorted(h for h in heads if h < last)
    blocks = []
    for a, b in zip(heads, heads[1:] + [last]):
        blocks.append((a, b))
    idx_of = {a: k for k, (a, b) in enumerate(blocks)}
    succ = defaultdict(list)
    for k, (a, b) in enumerate(blocks):
        term = None
        for i in range(b - 1, a - 1, -1):
            s = lines[i].strip()
            if s and not s.startswith(";") and not LABEL_RE.match(lines[i]):
                term = lines[i]
                break
        fall = True
        if term:
            m = BR_RE.match(term)
            if m:
                t = label_at.get(m.group(2))
                if t is not None:
                    succ[k].append(idx_of[t])
                if m.group(1) == "s_branch":
                    fall = False
            elif END_RE.match(term):
                fall = False
        if fall and k + 1 < len(blocks):
            succ[k].append(k + 1)

    def step(queue, i, reports):
        l = lines[i]
        m = WAIT_RE.match(l)
        if m:
            arg = m.group(1)
            n = None
            mm = re.search(r"lgkmcnt\((\d+)\)", arg)
            if mm:
                n = int(mm.group(1))
            else:
                mm = re.match(r"\s*(0x[0-9a-fA-F]+|\d+)\s*$", arg.split(";")[0])
                if mm:  # raw immediate: lgkmcnt is bits 11:8
                    n = (int(mm.group(1), 0) >> 8) & 15
            if n is not None and n < len(queue):
                if n == 0:
                    queue = ()
                elif any(q[1].startswith("s_") for q in queue):
                    # scalar-memory returns are OUT of order: with an s_load / s_buffer_load in the queue a
                    # counted wait proves nothing about the older LDS reads -- only lgkmcnt(0) retires them
                    pass
                else:
                    queue = queue[len(queue) - n:]
            return queue
        if l.strip().startswith("s_barrier"):
            pend = tuple(q for q in queue if q[1].startswith("ds_read") or q[1].startswith("ds_load"))
            if reports is not None and pend:
                reports[i] = pend
            return queue
        m = LGKM_RE.match(l)
        if m:
            queue = queue + ((i, m.group(1), l.strip()),)
            if len(queue) > CAP:
                # the counter saturates at CAP entries in the model: drop the oldest NON-read entries first, an
                # outstanding ds_read is never forgotten (it stays flagged until a wait retires it)
                keep = [q for q in queue if q[1].startswith("ds_read") or q[1].startswith("ds_load")]
                rest = [q for q in queue if not (q[1].startswith("ds_read") or q[1].startswith("ds_load"))]
                rest = rest[max(0, len(rest) - max(0, CAP - len(keep))):]
                queue = tuple(sorted(keep + rest))
        return queue

    inq = {0: ()}
    work = [0]
    seen_iter = 0
    while work and seen_iter < 200000:
        seen_iter += 1
        k = work.pop()
        q = inq[k]
        a, b = blocks[k]
        for i in range(a, b):
            q = step(q, i, None)
        for s in succ[k]:
            old = inq.get(s)
            nreads = lambda Q: sum(1 for e in Q if e[1].startswith("ds_read") or e[1].startswith("ds_load"))
            if old is None or (nreads(q), len(q)) > (nreads(old), len(old)):
                inq[s] = q
                work.append(s)
    reports = {}
    for k, (a, b) in enumerate(blocks):
        if k not in inq:
            continue
        q = inq[k]
        for i in range(a, b):
            q = step(q, i, reports)
    return reports


def first_use(lines, start, last, regs):
    for i in range(start, min(last, start + 400)):
        s = lines[i].split(";")[0]
        if not s.strip() or LABEL_RE.match(lines[i]):
            continue
        parts = s.strip().split(None, 1)
        if len(parts) < 2:
            continue
        ops = parts[1].split(",")
        srcs = ",".join(ops[1:]) if not parts[0].startswith(("ds_write", "ds_store", "global_store", "buffer_store")) else parts[1]
        if regs_of(srcs) & regs:
            return i
    return None


def next_lds_writers(lines, start, last, n=3):
    out = []
    for i in range(start, min(last, start + 600)):
        s = lines[i].strip()
        if s.startswith(("ds_write", "ds_store")) or ("global_load_lds" in s) or (s.startswith("buffer_load") and " lds" in s):
            out.append(i)
            if len(out) >= n:
                break
        if s.startswith("s_barrier") and i > start:
            break
    return out


def audit(path):
    """{mangled kernel name: (barriers, barriers with LDS reads possibly in flight)} of one listing"""
    lines = open(path).read().splitlines()
    out = {}
    for name, a, b in split_kernels(lines):
        nb = sum(1 for i in range(a, b) if lines[i].strip().startswith("s_barrier"))
        if nb:
            out[name] = (nb, len(analyse(lines, a, b)))
    return out


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    total = 0
    for name, a, b in split_kernels(lines):
        if flt and flt not in name:
            continue
        nb = sum(1 for i in range(a, b) if lines[i].strip().startswith("s_barrier"))
        if not nb:
            continue
        rep = analyse(lines, a, b)
        print("%s: %d barriers, %d with LDS reads possibly in flight" % (name[:150], nb, len(rep)))
        for i in sorted(rep):
            total += 1
            print("  barrier at line %d:" % (i + 1))
            for (j, op, text) in rep[i]:
                dst = regs_of(text.split(",")[0])
                u = first_use(lines, i + 1, b, dst)
                print("    line %d  %-60s first use behind the barrier: %s" % (j + 1, text[:60], ("line %d  %s" % (u + 1, lines[u].strip()[:70])) if u is not None else "none within 400 lines"))
            w = next_lds_writers(lines, i + 1, b)
            print("    next LDS writers behind it: %s" % (", ".join("line %d %s" % (x + 1, lines[x].strip().split()[0]) for x in w) or "none before the next barrier"))
    print("barriers with reads possibly in flight:", total)


if __name__ == "__main__":
    main()
