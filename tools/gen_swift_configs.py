"""Regenerate ska_sdp_exec_swiftly_amd/swift_configs.py (a compact table of the
parameter VALUES) from the reference catalogue.  Authoring container only."""
import importlib.util
import os

REF = "/root/reference/src/ska_sdp_exec_swiftly/swift_configs.py"
OUT = os.path.join(os.path.dirname(__file__), "..", "ska-sdp-distributed-fourier-transform_amd",
                   "ska_sdp_exec_swiftly_amd", "swift_configs.py")
FIELDS = ("W", "fov", "N", "Nx", "yB_size", "yN_size", "yP_size", "xA_size", "xM_size")


def main():
    spec = importlib.util.spec_from_file_location("ref_configs", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(OUT).read()
    head, rest = text.split('_TABLE = """\\\n', 1)
    tail = rest.split('"""', 1)[1]
    rows = "".join(
        " ".join([name] + [str(cfg[f]) for f in FIELDS]) + "\n" for name, cfg in mod.SWIFT_CONFIGS.items()
    )
    open(OUT, "w").write(head + '_TABLE = """\\\n' + rows + '"""' + tail)


if __name__ == "__main__":
    main()
