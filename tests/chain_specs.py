"""Hand-written test presets (our own text, same keys as the reference's .glslp files) and
the description the oracle chain needs for each shader identity."""
import os

# preset name -> (relative path below shaders_glsl/, text)
PRESETS = {
    "scanline": ("scanlines/scanline.glslp",
                 'shaders = "1"\nshader0 = "shaders/scanline.glsl"\n'),
    "crt-pi": ("crt/crt-pi.glslp",
               'shaders = "1"\nshader0 = "shaders/crt-pi.glsl"\nfilter_linear0 = "true"\n'
               'wrap_mode0 = "clamp_to_border"\nmipmap_input0 = "false"\nalias0 = ""\n'
               'float_framebuffer0 = "false"\nsrgb_framebuffer0 = "false"\n'),
    "stock": ("stock.glslp", 'shaders = "1"\nshader0 = "stock.glsl"\nfilter_linear0 = "false"\n'),
}

# shader identity -> oracle pass function, parameter order (name, default), extra samplers
SHADERS = {
    "stock.glsl": {"oracle": "stock", "params": [], "samplers": []},
    "scanlines/shaders/scanline.glsl": {
        "oracle": "scanline",
        "params": [("SCANLINE_BASE_BRIGHTNESS", 0.95), ("SCANLINE_SINE_COMP_A", 0.0), ("SCANLINE_SINE_COMP_B", 0.25),
                   ("size", 1.0)],
        "samplers": []},
    "crt/shaders/crt-pi.glsl": {
        "oracle": "crt_pi",
        "params": [("CURVATURE_X", 0.10), ("CURVATURE_Y", 0.15), ("MASK_BRIGHTNESS", 0.70), ("SCANLINE_WEIGHT", 6.0),
                   ("SCANLINE_GAP_BRIGHTNESS", 0.12), ("BLOOM_FACTOR", 1.5), ("INPUT_GAMMA", 2.4), ("OUTPUT_GAMMA", 2.2)],
        "samplers": []},
}


def write_tree(root):
    out = {}
    for name, (rel, text) in PRESETS.items():
        p = os.path.join(root, "shaders_glsl", rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)
        out[name] = p
    return out


def identity(shader_path):
    """Registry key of a resolved shader path: the part below shaders_glsl/, else the longest
    known identity that the path ends with (or that ends with the path's last two components)."""
    k = shader_path.rfind("shaders_glsl/")
    if k >= 0:
        return shader_path[k + len("shaders_glsl/"):]
    tail = "/".join(shader_path.split("/")[-2:])
    for ident in SHADERS:
        if shader_path.endswith("/" + ident) or ident.endswith("/" + tail) or ident == tail:
            return ident
    return shader_path
