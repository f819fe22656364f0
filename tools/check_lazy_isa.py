"""Build-time check of k_lz_match_lazy's hand-issued prefetch streams: tools/check_lazy_isa.py <asm file of zes_deflate.hip>

The loop's three requests per turn land in the accumulation registers a0 (W), a1 (D), a2 (S) and are fetched into vector
registers by the hand-written wait that covers them (zes_deflate.hip, LZ_REQ).  Nothing else in the kernel may name those
three registers — neither alone nor inside a register range — and every fetch must stand right behind an s_waitcnt vmcnt."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z15k_lz_match_lazy")][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
body = [l.split(";")[0].rstrip() for l in lines[start:end]]
req = re.compile(r"^\s*global_load_(ushort|dword) a([012]), v\d+, s\[\d+:\d+\]$")
fetch = re.compile(r"^\s*v_accvgpr_read_b32 v\d+, a([012])$")
bad = nreq = nfetch = 0
for i, t in enumerate(body):
    names = set(re.findall(r"\ba([012])\b", t))
    for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", t):
        names |= {str(n) for n in range(int(a), int(b) + 1) if n <= 2}
    if not names:
        continue
    if req.match(t):
        nreq += 1
        continue
    if fetch.match(t):
        nfetch += 1
        j = i - 1
        while j >= 0 and fetch.match(body[j]):
            j -= 1
        if not re.match(r"^\s*s_waitcnt vmcnt\(\d\)", body[j]):
            print("line %d: fetch without its wait: %s (behind: %s)" % (i, t.strip(), body[j].strip()))
            bad += 1
        continue
    print("line %d: a landing register is named outside a request or a fetch: %s" % (i, t.strip()))
    bad += 1
print("requests %d, fetches %d, violations %d" % (nreq, nfetch, bad))
sys.exit(1 if bad or nreq != nfetch or nreq < 18 or nreq % 3 else 0)  # three streams per instance of the loop (phases x forms x window maps)
