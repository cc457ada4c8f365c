python tools/pgs_stream_table.py --graphs 1024 --slots 0,256,512 --groups 2 --timeline 2>&1 | grep -v amdgpu.ids
