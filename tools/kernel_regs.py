#!/usr/bin/env python3
"""Print register / LDS / scratch use of every kernel in a hipcc -save-temps .s file."""
import re
import sys

s = open(sys.argv[1]).read()
pat = (r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)"
       r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)")
for m in re.finditer(pat, s, re.S):
    print(f"{m.group(3)[:60]:60s} vgpr {m.group(6):>3s} agpr {m.group(1):>3s} sgpr {m.group(5):>3s} lds {m.group(2):>6s} "
          f"scratch {m.group(4):>5s} spill {m.group(7)}")
