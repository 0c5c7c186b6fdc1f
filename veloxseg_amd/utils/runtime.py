"""Output-layout / loss-weight helpers on the hot path (reference: utils/runtime.py:125-144,158-174)."""


def normalized_deep_loss_weights(configured_weights, output_count):
    if output_count <= 0:
        raise ValueError("output_count must be greater than 0")
    weights = [float(w) for w in configured_weights]
    if not weights:
        raise ValueError("deep_Loss_weight must contain at least one value")
    total = sum(weights)
    if total == 0:
        raise ValueError("deep_Loss_weight sum must be non-zero")
    if len(weights) != output_count:
        if len(set(weights)) == 1:
            return [1.0 / output_count] * output_count
        raise ValueError("deep_Loss_weight length must match model deep-supervision outputs unless all configured weights are equal")
    return [w / total for w in weights]


def veloxseg_output_layout(output_count, num_modal):
    num_modal = int(num_modal)
    tail = 2 + num_modal
    if output_count <= tail:
        raise ValueError(f"VeloxSeg output count {output_count} is too small for {num_modal} modality reconstruction outputs")
    seg = output_count - tail
    return {"seg": (0, seg), "reconstruction": seg, "decoder_gram": seg + 1, "teacher_grams": tuple(range(seg + 2, seg + 2 + num_modal))}


def expected_input_channels(model_name, model_config):
    cfg = model_config.get(model_name)
    if cfg is None:
        return None
    in_ch = cfg.get("in_ch")
    if isinstance(in_ch, list):
        return sum(int(c) for c in in_ch)
    if isinstance(in_ch, int):
        return in_ch
    return None
