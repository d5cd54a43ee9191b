"""CPU ORACLE for the input pipeline -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/acr_oracle.py's header).

Restates myTool.py:1158-1199 (`get_data_from_chunk_v2`) and :1364-1403 (`get_data_from_chunk_val`) in numpy float64,
the precision the reference works in (`astype(np.float)` before every step), minus the decode: the functions take the
decoded RGB uint8 array that `cv2.imread` + `cv2.cvtColor(BGR2RGB)` produce.

Parity status: the geometry (RandomResizeLong :995-1008, flip :895-899, RandomCrop :923-955) and the normalisation
(:1180-1182) are literal restatements; the RESIZE is third-party arithmetic that is absent from /root/reference
(`opencv-python`, imported as cv2 at myTool.py:5; not installed in this image, not pinned in requirements.txt) -- PARITY
UNPINNED for that one step.  `cv2_resize_linear` restates OpenCV's published INTER_LINEAR rule for floating-point images
(modules/imgproc/src/resize.cpp, `resizeGeneric_` linear path): destination pixel d samples the source at
(d + 0.5) * src/dst - 0.5; a sample position left of pixel 0 reads pixel 0, one at or beyond the last pixel reads the
last pixel; no antialiasing.  The reference resizes float64 arrays, so OpenCV's fixed-point uint8 path does not apply.

The random draws are factored out: `draw_train_geometry` consumes the two generators in the reference's order
(np.random.uniform for flip_p at :1175, random.randint at :996, random.randrange at :935-945), so seeding
`random.Random` / `np.random.RandomState` reproduces a reference run whose globals were seeded the same way.
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def cv2_resize_linear(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) for a float (h, w, c) image, interpolation = INTER_LINEAR (the default)."""
    h, w = img.shape[:2]
    img = np.asarray(img, np.float64)

    def axis(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        lam = f - i0
        lam[i0 < 0] = 0.0
        i0[i0 < 0] = 0
        over = i0 >= n_src - 1
        lam[over] = 0.0
        i0[over] = n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, lam

    y0, y1, ly = axis(new_h, h)
    x0, x1, lx = axis(new_w, w)
    ly = ly[:, None, None]
    lx = lx[None, :, None]
    top = img[y0][:, x0] * (1 - lx) + img[y0][:, x1] * lx
    bot = img[y1][:, x0] * (1 - lx) + img[y1][:, x1] * lx
    return top * (1 - ly) + bot * ly


def resize_long_shape(h, w, target_long):
    """RandomResizeLong's target (myTool.py:999-1002) as the (width, height) tuple it hands to cv2.resize."""
    if w < h:
        return int(round(w * target_long / h)), target_long
    return target_long, int(round(h * target_long / w))


def random_crop_boxes(h, w, cropsize, pyrandom):
    """RandomCrop's draws (myTool.py:925-945), w before h."""
    ch, cw = min(cropsize, h), min(cropsize, w)
    w_space, h_space = w - cropsize, h - cropsize
    if w_space > 0:
        cont_left, img_left = 0, pyrandom.randrange(w_space + 1)
    else:
        cont_left, img_left = pyrandom.randrange(-w_space + 1), 0
    if h_space > 0:
        cont_top, img_top = 0, pyrandom.randrange(h_space + 1)
    else:
        cont_top, img_top = pyrandom.randrange(-h_space + 1), 0
    return dict(cont_top=cont_top, cont_left=cont_left, img_top=img_top, img_left=img_left, ch=ch, cw=cw)


def draw_train_geometry(h, w, dim, pyrandom, nprandom):
    """Per-image draws of get_data_from_chunk_v2 in the reference's order (:1175 flip_p, :996 target_long, :935-945 crop)."""
    flip_p = nprandom.uniform(0, 1)
    target_long = pyrandom.randint(int(dim * 0.9), int(dim / 0.875))
    new_w, new_h = resize_long_shape(h, w, target_long)
    g = dict(rw=new_w, rh=new_h, flip=int(flip_p > 0.5))
    g.update(random_crop_boxes(new_h, new_w, dim, pyrandom))
    return g


def normalise(img):
    out = np.array(img, np.float64)
    for c in range(3):
        out[:, :, c] = (out[:, :, c] / 255.0 - MEAN[c]) / STD[c]
    return out


def train_image(rgb_u8, dim, geom):
    """One image of get_data_from_chunk_v2 (:1176-1183) with the draws in `geom` -> (3, dim, dim) float32."""
    img = rgb_u8.astype(np.float64)
    img = cv2_resize_linear(img, geom["rw"], geom["rh"])
    if geom["flip"]:
        img = np.fliplr(img)
    img = normalise(img)
    box = np.zeros((dim, dim, 3), np.float32)
    ct, cl, it, il, ch, cw = (geom[k] for k in ("cont_top", "cont_left", "img_top", "img_left", "ch", "cw"))
    box[ct:ct + ch, cl:cl + cw] = img[it:it + ch, il:il + cw]
    return box.transpose(2, 0, 1)


def val_image(rgb_u8, dim):
    """One image of get_data_from_chunk_val (:1377-1387): plain resize to dim x dim, normalise."""
    img = cv2_resize_linear(rgb_u8.astype(np.float64), dim, dim)
    return normalise(img).astype(np.float32).transpose(2, 0, 1)


def get_data_from_chunk_v2(decoded, dim, pyrandom, nprandom):
    """myTool.py:1158-1199 over already decoded RGB uint8 arrays (the cv2.imread of :1176 is the caller's): the per-chunk
    `scale` draw of :1161 (never used, but it advances np.random), then per image the draws and the arithmetic above.
    Returns (images (B,3,dim,dim) float32, geometry list)."""
    nprandom.uniform(0.7, 1.3)
    out, geoms = [], []
    for rgb in decoded:
        g = draw_train_geometry(rgb.shape[0], rgb.shape[1], dim, pyrandom, nprandom)
        out.append(train_image(rgb, dim, g))
        geoms.append(g)
    return np.stack(out), geoms


def get_data_from_chunk_val(decoded, dim, nprandom):
    """myTool.py:1364-1403: `scale` draw (:1367), then resize to dim x dim + normalise per image."""
    nprandom.uniform(0.7, 1.3)
    return np.stack([val_image(rgb, dim) for rgb in decoded])
