import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"],3), "enter", round(d.get("enter_ms",0),3), "exit", round(d.get("exit_ms",0),3))
for k in d["roofline"]["per_class"]: print("   ", k["name"], k["launches_per_step"], round(k["event_ms_per_step"],3))
