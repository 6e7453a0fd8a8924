"""Fixture of the LONG instructions of the reference's evaluation annotations (run in the build container, where /root/reference exists):
every distinct instruction of /root/reference/lang_annotation_cache.json (1000 chains x 5 sub-tasks, the file eval_calvin.py feeds the
rollouts from) with more than 10 words - the ones that tokenize beyond 16 tokens once "<image>", "<|endofchunk|>" and eos are added
(data.py:905-919, max_length = 32) and that an 8-environment batch refused until round 4 - plus the word-count histogram of the whole file.
Instructions as the harness uses them: the first line of each entry (eval_utils.py:638-644).  Data only (instruction strings and counts); writes tests/golden/long_instructions.json."""
import collections
import json
import os

SRC = "/root/reference/lang_annotation_cache.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "long_instructions.json")

chains = json.load(open(SRC))
flat = [s.split("\n")[0] for chain in chains for s in chain]        # the harness feeds the first line (eval_utils.py:638-644)
hist = collections.Counter(len(s.split()) for s in flat)
long_ = sorted({s for s in flat if len(s.split()) > 10}, key=lambda s: (-len(s.split()), s))
json.dump({"source": "lang_annotation_cache.json (reference repo root)", "n_chains": len(chains), "n_instructions": len(flat),
           "word_count_histogram": {str(k): hist[k] for k in sorted(hist)}, "share_over_10_words": sum(v for k, v in hist.items() if k > 10) / len(flat),
           "long_instructions": long_}, open(OUT, "w"), indent=0)
print(len(long_), "distinct long instructions; share over 10 words", sum(v for k, v in hist.items() if k > 10) / len(flat), "max words", max(hist))
