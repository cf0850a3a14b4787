"""The rounds-1-4 VQA_* names of the A/B switches -> fields of vqa_index_options (include/vqa_retrieval.h).  The library reads no
environment any more; these dev scripts keep the old names as their command-line vocabulary and pass explicit options."""
ENV_TO_OPTION = {"VQA_STAGE_MIN": "stage_min_tiles", "VQA_STAGE_PCT": "stage_pct", "VQA_WIDE_K": "wide_k", "VQA_TWO_PASS": "two_pass",
                 "VQA_SEED_MULT": "seed_mult", "VQA_SEED_DIV": "seed_div", "VQA_SKETCH": "sketch", "VQA_SKETCH_CASCADE": "sketch_cascade",
                 "VQA_SKETCH_MID_K": "sketch_mid_k", "VQA_SKETCH_MID_MIN": "sketch_mid_min_tiles", "VQA_SKETCH_MID_PCT": "sketch_mid_pct",
                 "VQA_SKETCH_PRE_K": "sketch_pre_k", "VQA_POISON_WORKSPACE": "poison_workspace", "VQA_ONE_LAUNCH": "one_launch", "VQA_SKETCH_CENTER": "sketch_center",
                 "VQA_SKETCH_PER_ROW": "sketch_per_row", "VQA_SKETCH_ROTATE": "sketch_rotate", "VQA_SKETCH_COOLDOWN": "sketch_cooldown",
                 "VQA_SKETCH_PROFIT": "sketch_profit", "VQA_SKETCH_SPLIT": "sketch_split", "VQA_SKETCH_SX": "sketch_ring_stages",
                 "VQA_F16_LOOP": "f16_loop", "VQA_SKETCH_REGQ": "sketch_regq", "VQA_FINAL_RESCORE": "final_rescore", "VQA_RESCORE_COPY": "rescore_copy"}


def options(kv) -> dict:
    """{"VQA_STAGE_MIN": "2", ...} (or option field names) -> DeviceIndex(options=...)"""
    out = {}
    for k, v in dict(kv).items():
        key = ENV_TO_OPTION.get(k, k)
        out[key] = float(v) if key == "sketch_profit" else int(str(v), 0)
    return out
