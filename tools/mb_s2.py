"""Developer: the S2 drop-in seam measurement of bench.py alone (measure_s2_dropin), 32 pages.  python tools/mb_s2.py"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
from rapiddoc_amd.pages import synth_pages
from rapiddoc_amd.pipeline import PagePipelinePool, render_text_maps
states = bench.load_states()
pool = PagePipelinePool(states, rec_mode="strict", device=0, workers=1, rec_batch_num=64, rec_width_multiple=32, n_rec_streams=8, rec_chunking="adaptive")
pipe = pool.pipes[0]
pages_np, boxes = synth_pages(list(range(int(sys.argv[1]) if len(sys.argv) > 1 else 32)))
pages = torch.from_numpy(pages_np).cuda()
det_hw = pipe.det_forward(pages[:1])[1]
maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
print(json.dumps(bench.measure_s2_dropin(states, pipe, pages, maps, 0)))
