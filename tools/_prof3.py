import sys; sys.path.insert(0, "tools"); sys.argv=[sys.argv[0]]
exec(open("tools/_prof2.py").read().split("profile(1024, \"stretch_empty\", {\"escalate\": 0})")[0])
profile(1024, "stretch_empty", {"escalate": 0})
profile(1024, "stretch_empty", {"escalate": 0, "multiccd": 0})
