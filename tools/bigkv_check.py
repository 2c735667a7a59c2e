"""KV cache sized from free HBM (num_kvcache_blocks = -1) on Qwen3-8B shapes: prints the block count the
runner chose and serves a short batch from it. One MI355X: 6845 blocks = 1.75 M tokens = 240.6 GiB."""
import sys, os, time, json, torch
sys.path.insert(0, os.getcwd())
from tools.e2e_bench import hf_config
from light_vllm_amd.llm import LLM
from light_vllm_amd.sampling_params import SamplingParams
t0=time.time()
llm = LLM(None, hf_config=hf_config("8b"), enforce_eager=False, max_num_seqs=64, num_kvcache_blocks=-1,
          max_model_len=4096, random_seed=0, random_on_device=True)
free, total = torch.cuda.mem_get_info()
print(json.dumps({"kv_blocks": llm.config.num_kvcache_blocks, "kv_tokens": llm.config.num_kvcache_blocks*256,
                  "kv_GiB": round(llm.model_runner.kv_cache.numel()*2/2**30,1), "free_after_GiB": round(free/2**30,1),
                  "total_GiB": round(total/2**30,1), "init_s": round(time.time()-t0,1)}))
g = torch.Generator().manual_seed(0)
prompts = [torch.randint(0, 151936, (700,), generator=g).tolist() for _ in range(32)]
outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True))
print("completions", len(outs), llm.last_stats.get("avg_decode_tok_s"))
llm.exit()
