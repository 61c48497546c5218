import sys,json
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        j=json.loads(line); print(j["value"], "GCUPS", j["ms_per_step"], "ms/step; fwd", j["roofline"]["kernel_ms_avg"], j["checks"])
