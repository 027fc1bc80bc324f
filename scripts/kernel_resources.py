"""dev tool: VGPRs / spills / scratch / occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile).
  hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python scripts/kernel_resources.py res.txt [name filter]"""
import re, sys
cur, d = None, {}
for l in open(sys.argv[1]):
    m = re.search(r"remark: Function Name: (\S+)", l)
    if m:
        cur = m.group(1); d[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", l)
    if m and cur:
        d[cur][m.group(1).strip()] = int(m.group(2))
for k, v in d.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k[:60], {x: v.get(x) for x in ("VGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy")})
