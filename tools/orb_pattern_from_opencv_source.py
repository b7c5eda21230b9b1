"""Extract OpenCV's learned rBRIEF sampling table from an OpenCV SOURCE TREE into the text file the ORB branch reads
(UVO_ORB_PATTERN_FILE / uvo_hip::set_orb_pattern / uvo_orb_set_pattern):

    python tools/orb_pattern_from_opencv_source.py /path/to/opencv/modules/features2d/src/orb.cpp bit_pattern_31.txt

The table (`static int bit_pattern_31_[256*4]`, 1024 integers learned offline: x0, y0, x1, y1 per descriptor bit) is OpenCV's data; this
repository neither carries nor restates it.  The script reads the array initialiser, drops the comments, checks that there are 1024
integers within the 31 x 31 patch and writes them four per line."""
import re
import sys


def extract(text: str, name: str = "bit_pattern_31_"):
    m = re.search(r"\b" + re.escape(name) + r"\s*\[[^\]]*\]\s*=\s*\{", text)
    if not m:
        raise ValueError(f"no initialiser of {name}[] in the source")
    depth, i = 1, m.end()
    while i < len(text) and depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    if depth:
        raise ValueError("unterminated initialiser")
    body = text[m.end():i - 1]
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)          # /* mean (0), correlation (0) */ annotations
    body = re.sub(r"//[^\n]*", " ", body)
    vals = [int(v) for v in re.findall(r"[-+]?\d+", body)]
    if len(vals) != 1024:
        raise ValueError(f"expected 1024 integers, found {len(vals)}")
    if max(abs(v) for v in vals) > 15:
        raise ValueError("a coordinate lies outside the 31 x 31 patch")
    return vals


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    vals = extract(open(argv[1], errors="replace").read())
    with open(argv[2], "w") as f:
        for k in range(256):
            f.write(", ".join(str(v) for v in vals[4 * k:4 * k + 4]) + (",\n" if k < 255 else "\n"))
    print(f"wrote 256 test pairs to {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
