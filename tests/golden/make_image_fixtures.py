#!/usr/bin/env python3
"""Makes the BMP / TGA / PNG texture fixtures (tests/golden/assets/images/*) and their reference decodes (tests/golden/ref_stb_decode_images.json).

Every file is written byte by byte here (no imaging library), so each header variant the decoders have to know is present on purpose.  The
reference decodes come from the REFERENCE'S OWN stb_image build: oracle/_ref/ref_host `decode <file>` calls stbi_load(path, &w, &h, &n, 3)
exactly as src/gpu_scene_builder.cpp:215 does.  Run in the build container only (needs /root/reference to have built oracle/_ref/ref_host);
the outputs are committed, the reference is not."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "assets", "images")
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_host")


def picture(w, h, seed, channels=3):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.float64)
    img[..., 0] = 128 + 100 * np.sin(x / 3.0) * np.cos(y / 4.0)
    img[..., 1] = (x * 255.0 / max(1, w - 1) + y * 64.0 / max(1, h - 1)) % 256
    img[..., 2] = 255 * ((x // 2 + y // 3) % 2)
    img[..., 3] = (x * 37 + y * 91) % 256
    img[..., :3] += rng.normal(0, 10, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)[..., :channels]


# ---------------------------------------------------------------- BMP
def bmp_rows(rows, bottom_up):
    rows = [r + b"\0" * ((-len(r)) & 3) for r in rows]
    return b"".join(reversed(rows) if bottom_up else rows)


def bmp_file(info_header, palette, pixels, extra_after_header=b""):
    offset = 14 + len(info_header) + len(extra_after_header) + len(palette)
    head = b"BM" + struct.pack("<IHHI", offset + len(pixels), 0, 0, offset)
    return head + info_header + extra_after_header + palette + pixels


def info40(w, h, bpp, compress=0, size=40, masks=b"", colours=0):
    base = struct.pack("<IiiHHIIiiII", size, w, h, 1, bpp, compress, 0, 2835, 2835, colours, 0)
    return base + masks + b"\0" * (size - 40 - len(masks))


def pack_bits(indices, bits):
    out = bytearray()
    acc, n = 0, 0
    for v in indices:
        acc = (acc << bits) | int(v)
        n += bits
        if n == 8:
            out.append(acc)
            acc, n = 0, 0
    if n:
        out.append(acc << (8 - n))
    return bytes(out)


def make_bmps():
    files = {}
    rgb = picture(13, 7, 1)
    bgr_rows = [rgb[y, :, ::-1].tobytes() for y in range(7)]
    files["bmp24_bottom_up"] = bmp_file(info40(13, 7, 24), b"", bmp_rows(bgr_rows, True))
    files["bmp24_top_down"] = bmp_file(info40(13, -7, 24), b"", bmp_rows(bgr_rows, False))
    files["bmp24_os2_header"] = bmp_file(struct.pack("<IHHHH", 12, 13, 7, 1, 24), b"", bmp_rows(bgr_rows, True))
    files["bmp24_v4_header"] = bmp_file(info40(13, 7, 24, size=108), b"", bmp_rows(bgr_rows, True))
    files["bmp24_gap_before_pixels"] = bmp_file(info40(13, 7, 24), b"", bmp_rows(bgr_rows, True), extra_after_header=b"\x55" * 20)
    rgba = picture(10, 6, 2, 4)
    bgra_rows = [rgba[y][:, [2, 1, 0, 3]].tobytes() for y in range(6)]
    files["bmp32_plain"] = bmp_file(info40(10, 6, 32), b"", bmp_rows(bgra_rows, True))
    files["bmp32_v5_bitfields_alpha"] = bmp_file(info40(10, 6, 32, compress=3, size=124, masks=struct.pack("<IIII", 0xFF0000, 0xFF00, 0xFF, 0xFF000000)), b"",
                                                 bmp_rows(bgra_rows, True))
    # 32 bits with odd masks: 10-bit fields are refused, 3 / 6 / 7-bit fields are widened by bit replication
    v = (rgba[..., 0].astype(np.uint32) >> 5) << 29 | (rgba[..., 1].astype(np.uint32) >> 2) << 12 | (rgba[..., 2].astype(np.uint32) >> 1) << 1
    rows = [v[y].astype("<u4").tobytes() for y in range(6)]
    files["bmp32_bitfields_3_6_7"] = bmp_file(info40(10, 6, 32, compress=3), b"", bmp_rows(rows, True), extra_after_header=struct.pack("<III", 7 << 29, 63 << 12, 127 << 1))
    files["bmp32_bitfields_10bit_refused"] = bmp_file(info40(10, 6, 32, compress=3), b"", bmp_rows(rows, True), extra_after_header=struct.pack("<III", 1023 << 20, 1023 << 10, 1023))
    p = picture(9, 5, 3)
    v555 = (p[..., 0].astype(np.uint16) >> 3) << 10 | (p[..., 1].astype(np.uint16) >> 3) << 5 | (p[..., 2].astype(np.uint16) >> 3)
    v565 = (p[..., 0].astype(np.uint16) >> 3) << 11 | (p[..., 1].astype(np.uint16) >> 2) << 5 | (p[..., 2].astype(np.uint16) >> 3)
    files["bmp16_555"] = bmp_file(info40(9, 5, 16), b"", bmp_rows([v555[y].astype("<u2").tobytes() for y in range(5)], True))
    files["bmp16_565_bitfields"] = bmp_file(info40(9, 5, 16, compress=3), b"", bmp_rows([v565[y].astype("<u2").tobytes() for y in range(5)], True),
                                            extra_after_header=struct.pack("<III", 0xF800, 0x07E0, 0x001F))
    rng = np.random.default_rng(4)
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    pal4 = b"".join(bytes([c[2], c[1], c[0], 0]) for c in pal)
    idx8 = rng.integers(0, 256, (6, 11), dtype=np.uint8)
    files["bmp8_palette"] = bmp_file(info40(11, 6, 8), pal4, bmp_rows([idx8[y].tobytes() for y in range(6)], True))
    idx8s = rng.integers(0, 40, (6, 11), dtype=np.uint8)
    files["bmp8_palette_40_entries"] = bmp_file(info40(11, 6, 8, colours=40), pal4[:160], bmp_rows([idx8s[y].tobytes() for y in range(6)], True))
    # (with the 12-byte header the reference reads (offset - 14 - 24) / 3 = 252 of the 256 entries and leaves the rest uninitialised:
    #  indices stay below 252 here)
    files["bmp8_os2_palette"] = bmp_file(struct.pack("<IHHHH", 12, 11, 6, 1, 8), b"".join(bytes([c[2], c[1], c[0]]) for c in pal),
                                         bmp_rows([np.minimum(idx8[y], 251).tobytes() for y in range(6)], True))
    idx4 = rng.integers(0, 16, (5, 7), dtype=np.uint8)
    files["bmp4_palette"] = bmp_file(info40(7, 5, 4), pal4[:64], bmp_rows([pack_bits(idx4[y], 4) for y in range(5)], True))
    idx1 = rng.integers(0, 2, (4, 19), dtype=np.uint8)
    files["bmp1_palette"] = bmp_file(info40(19, 4, 1), pal4[:8], bmp_rows([pack_bits(idx1[y], 1) for y in range(4)], True))
    files["bmp8_rle_refused"] = bmp_file(info40(11, 6, 8, compress=1), pal4, b"\x0b\x05\x00\x00" * 6 + b"\x00\x01")
    return {k + ".bmp": v for k, v in files.items()}


# ---------------------------------------------------------------- TGA
def tga_header(id_len, map_type, img_type, map_start, map_len, map_bits, w, h, bits, descriptor):
    return struct.pack("<BBBHHBHHHHBB", id_len, map_type, img_type, map_start, map_len, map_bits, 0, 0, w, h, bits, descriptor)


def rle_encode(pixels):
    """pixels: list of bytes objects (one per pixel).  Mixes run packets and raw packets."""
    out = bytearray()
    i = 0
    n = len(pixels)
    while i < n:
        run = 1
        while i + run < n and run < 128 and pixels[i + run] == pixels[i]:
            run += 1
        if run >= 2:
            out.append(0x80 | (run - 1))
            out += pixels[i]
            i += run
        else:
            j = i + 1
            while j < n and j - i < 128 and (j + 1 >= n or pixels[j] != pixels[j + 1]):
                j += 1
            out.append(j - i - 1)
            for k in range(i, j):
                out += pixels[k]
            i = j
    return bytes(out)


def make_tgas():
    files = {}
    rgb = picture(12, 7, 11)
    rgb[2, 3:9] = rgb[2, 3]                                      # runs for the RLE variants
    rgb[5, :] = rgb[5, 0]
    bgr = rgb[..., ::-1]
    files["tga24_bottom_up"] = tga_header(0, 0, 2, 0, 0, 0, 12, 7, 24, 0) + bgr[::-1].tobytes()
    files["tga24_top_down_with_id"] = tga_header(5, 0, 2, 0, 0, 0, 12, 7, 24, 0x20) + b"hello" + bgr.tobytes()
    rgba = picture(12, 7, 12, 4)
    bgra = rgba[..., [2, 1, 0, 3]]
    files["tga32_bottom_up"] = tga_header(0, 0, 2, 0, 0, 0, 12, 7, 32, 8) + bgra[::-1].tobytes()
    v555 = ((rgb[..., 0].astype(np.uint16) >> 3) << 10 | (rgb[..., 1].astype(np.uint16) >> 3) << 5 | (rgb[..., 2].astype(np.uint16) >> 3)).astype("<u2")
    files["tga16_555"] = tga_header(0, 0, 2, 0, 0, 0, 12, 7, 16, 0) + v555[::-1].tobytes()
    files["tga15_555_top_down"] = tga_header(0, 0, 2, 0, 0, 0, 12, 7, 15, 0x20) + v555.tobytes()
    gray = picture(9, 6, 13)[..., 0].copy()
    gray[3, 2:8] = 77
    files["tga8_gray"] = tga_header(0, 0, 3, 0, 0, 0, 9, 6, 8, 0) + gray[::-1].tobytes()
    ga = np.stack([gray, 255 - gray], axis=-1)
    files["tga16_gray_alpha"] = tga_header(0, 0, 3, 0, 0, 0, 9, 6, 16, 8) + ga[::-1].tobytes()
    rng = np.random.default_rng(14)
    pal = rng.integers(0, 256, (50, 3), dtype=np.uint8)
    idx = rng.integers(0, 50, (6, 9), dtype=np.uint8)
    idx[1, 1:7] = 9
    files["tga_indexed_24bit_map"] = tga_header(0, 1, 1, 0, 50, 24, 9, 6, 8, 0) + pal[:, ::-1].tobytes() + idx[::-1].tobytes()
    pal16 = rng.integers(0, 1 << 15, 50, dtype=np.uint16).astype("<u2")
    files["tga_indexed_16bit_map"] = tga_header(0, 1, 1, 0, 50, 16, 9, 6, 8, 0x20) + pal16.tobytes() + idx.tobytes()
    pal32 = rng.integers(0, 256, (50, 4), dtype=np.uint8)
    files["tga_indexed_32bit_map_16bit_indices"] = tga_header(0, 1, 1, 0, 50, 32, 9, 6, 16, 0) + pal32.tobytes() + idx[::-1].astype("<u2").tobytes()
    files["tga_indexed_index_out_of_range"] = tga_header(0, 1, 1, 0, 20, 24, 9, 6, 8, 0) + pal[:20, ::-1].tobytes() + idx[::-1].tobytes()
    files["tga24_rle"] = tga_header(0, 0, 10, 0, 0, 0, 12, 7, 24, 0) + rle_encode([bytes(px) for px in bgr[::-1].reshape(-1, 3)])
    files["tga32_rle_top_down"] = tga_header(0, 0, 10, 0, 0, 0, 12, 7, 32, 0x28) + rle_encode([bytes(px) for px in bgra.reshape(-1, 4)])
    files["tga16_rle"] = tga_header(0, 0, 10, 0, 0, 0, 12, 7, 16, 0) + rle_encode([bytes(px) for px in v555[::-1].reshape(-1, 1).view(np.uint8).reshape(-1, 2)])
    files["tga8_gray_rle"] = tga_header(0, 0, 11, 0, 0, 0, 9, 6, 8, 0) + rle_encode([bytes([px]) for px in gray[::-1].reshape(-1)])
    files["tga_indexed_rle"] = tga_header(0, 1, 9, 0, 50, 24, 9, 6, 8, 0) + pal[:, ::-1].tobytes() + rle_encode([bytes([px]) for px in idx[::-1].reshape(-1)])
    return {k + ".tga": v for k, v in files.items()}


# ---------------------------------------------------------------- PNG
def png_chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, first_filter):
    """rows: list of bytes.  Filter type cycles 0..4 from first_filter, so every type and its first-row special case appears."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for k, row in enumerate(rows):
        ft = (first_filter + k) % 5
        out.append(ft)
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) // 2, paeth(a, b, c)][ft]
            out.append((x - pred) & 0xFF)
        prev = row
    return bytes(out)


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def png_file(samples, depth, ctype, interlace, palette=None, first_filter=0):
    """samples: H x W x C array of integer sample values (already in the range of `depth`)."""
    h, w, c = samples.shape

    def pack_row(px):                                             # px: W' x C
        if depth == 8:
            return px.astype(np.uint8).tobytes()
        if depth == 16:
            return px.astype(">u2").tobytes()
        return pack_bits(px.reshape(-1), depth)
    bpp = max(1, c * depth // 8)
    raw = b""
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    for n, (x0, y0, dx, dy) in enumerate(passes):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += filter_rows([pack_row(sub[y]) for y in range(sub.shape[0])], bpp, first_filter + n)
    body = png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        body += png_chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    comp = zlib.compress(raw, 9)
    half = len(comp) // 2
    body += png_chunk(b"IDAT", comp[:half]) + png_chunk(b"IDAT", comp[half:]) + png_chunk(b"IEND", b"")
    return b"\x89PNG\r\n\x1a\n" + body


def make_pngs():
    files = {}
    rng = np.random.default_rng(21)
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    for tag, interlace in (("plain", False), ("adam7", True)):
        files[f"png_rgb8_{tag}_17x13"] = png_file(picture(17, 13, 22), 8, 2, interlace, first_filter=1)
        files[f"png_rgba8_{tag}_9x9"] = png_file(picture(9, 9, 23, 4), 8, 6, interlace, first_filter=2)
        files[f"png_rgb16_{tag}_5x6"] = png_file(picture(5, 6, 24).astype(np.uint16) * 257 + 13, 16, 2, interlace, first_filter=3)
        files[f"png_gray8_{tag}_10x3"] = png_file(picture(10, 3, 25)[..., :1], 8, 0, interlace, first_filter=4)
        files[f"png_gray16_{tag}_4x4"] = png_file(picture(4, 4, 26)[..., :1].astype(np.uint16) * 201 + 7, 16, 0, interlace)
        files[f"png_grayalpha8_{tag}_7x5"] = png_file(picture(7, 5, 27, 4)[..., [0, 3]], 8, 4, interlace, first_filter=1)
        for bits in (1, 2, 4):
            files[f"png_gray{bits}_{tag}_13x6"] = png_file(rng.integers(0, 1 << bits, (6, 13, 1)), bits, 0, interlace, first_filter=bits)
            files[f"png_palette{bits}_{tag}_11x7"] = png_file(rng.integers(0, 1 << bits, (7, 11, 1)), bits, 3, interlace, palette=pal[:1 << bits], first_filter=bits + 1)
        files[f"png_palette8_{tag}_6x9"] = png_file(rng.integers(0, 200, (9, 6, 1)), 8, 3, interlace, palette=pal[:200], first_filter=2)
    for w, h in ((1, 1), (2, 1), (1, 3), (3, 2), (5, 1), (8, 8)):          # interlaced images some of whose passes are empty
        files[f"png_rgb8_adam7_{w}x{h}"] = png_file(picture(w, h, 30 + w * 9 + h), 8, 2, True, first_filter=w + h)
    return {k + ".png": v for k, v in files.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = {}
    for name, data in {**make_bmps(), **make_tgas(), **make_pngs()}.items():
        path = os.path.join(OUT, name)
        with open(path, "wb") as f:
            f.write(data)
        r = subprocess.run([REF, "decode", path], capture_output=True, text=True)
        ref[name] = json.loads(r.stdout) if r.stdout.strip().startswith("{") else {"ok": 0}
    with open(os.path.join(HERE, "ref_stb_decode_images.json"), "w") as f:
        json.dump(ref, f, indent=0, sort_keys=True)
    ok = sum(1 for v in ref.values() if v.get("ok"))
    print(f"{len(ref)} files, {ok} decoded by the reference, {len(ref) - ok} refused: {[k for k, v in ref.items() if not v.get('ok')]}")


if __name__ == "__main__":
    main()
