"""Generates the device-side product schedules of mul_wide / sqr_wide in bulletproofs_amd/csrc/fe.h.

Two product-scanning chains (low columns / high columns) advance one product each in turn; a run of
products without a column end in either chain becomes one asm block (so the compiler's inter-asm
hazard padding, one s_nop per adjacent pair of asm statements on gfx950, never triggers: between two
blocks there is always the accumulator rotation of the chain whose column ended).
Usage: python benchmarks/gen_fe_schedule.py mul|sqr   -> C++ statements on stdout."""
import sys


def sequences(kind):
    low, high = [], []
    if kind == "mul":
        for s in range(8):
            for q in range(s + 1):
                low.append((("a", q), ("b", s - q), q == 0, q == s, s))
        for s in range(7):
            n = 7 - s
            for q in range(n):
                high.append((("a", s + 1 + q), ("b", 7 - q), q == 0, q == n - 1, 8 + s))
    else:
        for s in range(1, 8):
            prs = [(q, s - q) for q in range(4) if q < s - q]
            for idx, (i, j) in enumerate(prs):
                low.append((("a", i), ("a", j), idx == 0, idx == len(prs) - 1, s))
        for s in range(1, 7):
            prs = [(s + q, 7 - q) for q in range(4) if s + q < 7 - q]
            for idx, (i, j) in enumerate(prs):
                high.append((("a", i), ("a", j), idx == 0, idx == len(prs) - 1, 7 + s))
    return low, high


def main(kind):
    low, high = sequences(kind)
    out_arr = "t" if kind == "mul" else "o"
    merged = []
    for n in range(max(len(low), len(high))):
        if n < len(low):
            merged.append(("L", low[n]))
        if n < len(high):
            merged.append(("H", high[n]))
    blocks, cur = [], []
    for ch, e in merged:
        cur.append((ch, e))
        if e[3]:  # column end: close the block
            blocks.append(cur)
            cur = []
    assert not cur
    lines = []
    for blk in blocks:
        ops_out, ops_in, text = [], [], []
        idx = {}

        def out_op(name, constraint):
            if name not in idx:
                idx[name] = None
                ops_out.append((name, constraint))
        # outputs first: acc/ext of the chains present
        first_seen = {}
        for ch, e in blk:
            if ch not in first_seen:
                first_seen[ch] = e[2]  # is the chain's first product in this block a column start?
        for ch in first_seen:
            out_op("acc" + ch, "+v")
            out_op("ext" + ch, "=&v" if first_seen[ch] else "+v")
        regs = []
        for ch, e in blk:
            for arr, i in (e[0], e[1]):
                nm = "%s.v[%d]" % (arr, i)
                if nm not in regs:
                    regs.append(nm)
        names = [n for n, _ in ops_out] + regs
        num = {n: k for k, n in enumerate(names)}
        for ch, e in blk:
            x = "%%%d" % num["%s.v[%d]" % e[0]]
            y = "%%%d" % num["%s.v[%d]" % e[1]]
            acc = "%%%d" % num["acc" + ch]
            ext = "%%%d" % num["ext" + ch]
            text.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, x, y, acc))
            if e[2]:
                text.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % ext)
            else:
                text.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (ext, ext))
        asm = '  asm("' + '\\n\\t"\n      "'.join(text) + '"\n      : ' + ", ".join('"%s"(%s)' % (c, n) for n, c in ops_out)
        asm += "\n      : " + ", ".join('"v"(%s)' % r for r in regs) + '\n      : "vcc");'
        lines.append(asm)
        ch, e = blk[-1]
        lines.append("  %s[%d] = (uint32_t)acc%s;" % (out_arr, e[4], ch))
        lines.append("  acc%s = (acc%s >> 32) | ((uint64_t)ext%s << 32);" % (ch, ch, ch))
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1])
