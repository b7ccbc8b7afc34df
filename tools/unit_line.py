import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
u=d.get('keyframe_unit',{})
print({k:u[k] for k in u if k not in ('note','moved')})
print('moved', {k:v for k,v in u.get('moved',{}).items() if k!='note'})
print('step', d['ms_per_step'], d['resident']['ms_per_step'])
