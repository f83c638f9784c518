import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.helpers import load_hr_case, hr_roles
from latte_amd.hr_module import HR_Block
z,cfg,batch = load_hr_case('w5_360')
ab,at,hb,ht,a = hr_roles(z,batch,'23')
ab = dict(ab, seg_logit=a.cuda()); hb = dict(hb, seg_logit=hb['seg_logit'].cuda())
d = HR_Block(cfg["temp_wd_size"], cfg["max_range"], cfg["voxel_size"], cfg["conf_perc"])(ab,at,hb,ht)
for i in range(5):
    got = d['ac_voxels'][i].cpu().numpy(); want = z[f'23_ac_voxels_{i}']
    gs = set(map(tuple, got)); ws = set(map(tuple, want))
    print(i, got.shape, want.shape, 'missing', sorted(ws-gs)[:5], 'extra', sorted(gs-ws)[:5])
# analyze anchor 4: voxels from anchor points
i=4
p = ab['ori_points'][i].numpy()
vs=0.2; mo=-251
c = np.floor(p.astype(np.float64)/vs).astype(np.int32)-mo
want = z[f'23_ac_voxels_{i}']; got = d['ac_voxels'][i].cpu().numpy()
ws = set(map(tuple,want)); gs=set(map(tuple,got))
for v in sorted(ws-gs):
    m = (c==np.array(v)).all(1)
    print('missing voxel', v, 'anchor pts', p[m], 'norm', np.linalg.norm(p[m],axis=1))
    # hist points in that voxel
    for j in (2,3,4):
        T = torch.linalg.inv(ab['pose'][i]) @ hb['pose'][j]
        q = hb['ori_points'][j]
        qh = torch.cat([q, torch.ones(q.shape[0],1)],1).double()
        t = (T @ qh.T).T[:, :3].numpy()
        ch = np.floor(t/vs).astype(np.int32)-mo
        mm = (ch==np.array(v)).all(1)
        if mm.any(): print('   hist frame', j, t[mm], np.linalg.norm(t[mm],axis=1))
