"""Extracts the reference's static (vk, proof, inputs) triple -- DATA, read as text -- from
/root/reference/test/test_verify.py:10-12 into tests/golden/ref_static_triple.json.
Run once in the build container (the reference does not travel to the GPU box)."""
import ast, json, re, sys, pathlib
src = pathlib.Path("/root/reference/test/test_verify.py").read_text()
vk = ast.literal_eval(re.search(r"^VK_STATIC = (\{.*\})$", src, re.M).group(1))
proof = ast.literal_eval(re.search(r"^PROOF_STATIC = (\{.*\})$", src, re.M).group(1))
out = pathlib.Path(__file__).with_name("ref_static_triple.json")
out.write_text(json.dumps({"source": "test/test_verify.py:10-12", "vk": vk, "proof": proof}, indent=1) + "\n")
print("wrote", out)
