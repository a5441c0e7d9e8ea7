import sys, ast
rows=[ast.literal_eval(l) for l in sys.stdin if l.startswith("{'step'")]
def m(a,b): v=[r['wall_ms'] for r in rows if a<=r['step']<=b]; return round(sum(v)/len(v),3) if v else None
print("4-23", m(4,23), "24-43", m(24,43), "100-119", m(100,119), "4-119", m(4,119))
