set -x
mkdir -p gpurun_out
(clinfo 2>&1 | grep -E 'Number of platforms|Platform Name|Device Name|Number of devices|Driver Version' | head -10) > gpurun_out/clinfo.txt 2>&1
rocminfo | grep -E 'Marketing Name|Compute Unit|Max Clock' | head -8 >> gpurun_out/clinfo.txt
nproc >> gpurun_out/clinfo.txt; lscpu | grep 'Model name' >> gpurun_out/clinfo.txt
timeout 600 python tests/make_ref_fixtures.py > gpurun_out/ref_fixtures.log 2>&1; echo "fixtures rc=$?" >> gpurun_out/ref_fixtures.log
timeout 300 python - > gpurun_out/hip_smoke.log 2>&1 <<'PY'
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import voxel_raycaster_amd as vrc
from oracle import orc
import scenes
atlas=scenes.hash_atlas()
for f in scenes.ALL:
    s=f(); dim=s['dim']
    m=vrc.Map(dim, s['grid'], buffer_size=100000)
    for using in (1,0):
        c=vrc.CLCaster(); assert c.init(0), "init"
        c.add_to_settings_buffer("octree_dimensions","OCTDIM",dim)
        c.add_to_settings_buffer("using_octree","OCTENABLED",using)
        md=20 if dim<=16 else 3*dim
        c.add_to_settings_buffer("max_distance","MAX_DISTANCE",md)
        assert c.assign_octree(m) and c.assign_map(m)
        cd=np.array(s['cam_dir'],dtype=np.float32); cp=np.array(s['cam_pos'],dtype=np.float32)
        assert c.assign_camera(cd,cp)
        assert c.create_viewport(160,120)
        li=np.zeros((8,10),np.float32); li[:1]=s['lights']
        assert c.assign_lights(li)
        assert c.create_texture_atlas(atlas,(16,16))
        assert c.validate(), c.last_error()
        assert c.compute(), c.last_error()
        img=c.read_image(); hits=c.read_hits(); ctr=c.counters()
        oimg,ohits,octr=orc.raycast(width=160,height=120,cam_dir=s['cam_dir'],cam_pos=s['cam_pos'],lights=li,atlas=atlas,tile_dim=(16,16),descriptors=m.octree.descriptor_buffer,root_index=m.octree.root_index,octree_dim=dim,using_octree=using,grid=s['grid'],max_distance=md)
        print(s['name'],'using',using,'img bit-equal',np.array_equal(img.view(np.uint32),oimg.view(np.uint32)),'maxabs',float(np.abs(img-oimg).max()),'hits equal',np.array_equal(hits,ohits), 'ctr', ctr, 'octr', octr, c.timing())
PY
echo done
