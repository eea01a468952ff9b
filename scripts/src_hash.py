#!/usr/bin/env python3
"""The sources a kernel is built from, as one hash: the .hip file(s) named + every header of mgl_amd/csrc/ they include, transitively.

    python scripts/src_hash.py sw_dp16_lane_ck.hip            # prints the hash
    python scripts/src_hash.py --stamp                         # writes it into every profiles/pmc_traffic.json entry that names `sources`
                                                               # and carries `"restamp": true` (an entry just measured on THIS tree)

profiles/pmc_traffic.json entries carry `sources` (the .hip files of their kernel) and `src_hash`; bench.py recomputes the hash when it
runs (no git needed on the GPU box) and prints an entry's counters in the line only when they were taken on the sources it is running --
a stale entry becomes `null` with the reason, never a number that looks current (round 4's review, item 8)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mgl_amd", "csrc")
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def closure(files):
    """The named files of mgl_amd/csrc/ + every local header they include, transitively; sorted base names."""
    seen, todo = set(), [os.path.basename(f) for f in files]
    while todo:
        f = todo.pop()
        path = os.path.join(CSRC, f)
        if f in seen or not os.path.exists(path):
            continue
        seen.add(f)
        for inc in _INC.findall(open(path, encoding="utf-8", errors="replace").read()):
            if os.path.exists(os.path.join(CSRC, os.path.basename(inc))) and "/" not in inc.replace("./", ""):
                todo.append(os.path.basename(inc))
    return sorted(seen)


def source_hash(files):
    """sha256 over (name, bytes) of closure(files), first 16 hex digits; None when a named file is missing."""
    names = closure(files)
    if not names or any(os.path.basename(f) not in names for f in files):
        return None
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0")
        h.update(open(os.path.join(CSRC, n), "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def check(entry):
    """(ok, reason) for a pmc_traffic.json entry against the tree this runs in."""
    src = entry.get("sources")
    if not src or not entry.get("src_hash"):
        return False, "the entry records no source hash (taken before round 5): not known to be of this build"
    now = source_hash(src)
    if now != entry["src_hash"]:
        return False, f"stale: taken on sources {entry['src_hash']} (commit {entry.get('commit', '?')}), this build's {'+'.join(src)} hash to {now}"
    return True, f"sources {now} (commit {entry.get('commit', '?')})"


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--stamp":
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        rec = json.load(open(path))
        n = 0
        for kernel, entries in rec.items():
            if not isinstance(entries, list):
                continue
            for e in entries:
                if e.pop("restamp", False) and e.get("sources"):
                    e["src_hash"] = source_hash(e["sources"])
                    n += 1
        json.dump(rec, open(path, "w"), indent=1)
        print(f"stamped {n} entries")
        return
    print(source_hash(sys.argv[1:]), " ".join(closure(sys.argv[1:])))


if __name__ == "__main__":
    main()
