"""``GeoDataFrame.to_crs`` for the outline layers (reference helpers.py:55, 789-790 and preprocessing.py:158 reproject the
forest outline / exclude files to the rasters' CRS with geopandas → pyproj → PROJ; none of them is installed here).

What is restated: the coordinate operations between the systems such files come in — geographic WGS 84 / ETRS89
(EPSG:4326, 4258, CRS84), their UTM zones (EPSG:326zz / 327zz, ETRS89 / UTM 258zz for zones 28 - 38; the reference's rasters
are EPSG:25832) and Web Mercator (EPSG:3857) — vertex by vertex, as ``to_crs`` does (no densification). Transverse Mercator
= the Krüger series in the third flattening to order n^6 (Karney 2011, "Transverse Mercator with an accuracy of a few
nanometers", eqs. 35 / 36 with the alpha / beta coefficients PROJ's ``etmerc`` / ``tmerc`` uses): round trip < 1e-6 m inside
a zone. WGS 84 and ETRS89 are treated as the same datum (they differ by the plate motion since 1989, < 1 m, and PROJ's
default pipeline between EPSG:4326 and EPSG:25832 is the same null transformation). Any other pair of codes raises — as
before this module existed — instead of guessing. Parity: unpinned (no PROJ here); tests/test_crs.py checks the series against
a numerical quadrature of the meridian arc, Snyder's published worked example and round trips.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

A_GRS80 = 6378137.0
F_GRS80 = 1.0 / 298.257222101          # WGS 84: 1 / 298.257223563 (0.1 mm at the pole)
K0_UTM = 0.9996
GEOGRAPHIC = {4326, 4258, 4979, 4937}


def _series(n: float):
    """Rectifying radius A and the alpha (forward) / beta (inverse) coefficients of the Krüger series, order n^6."""
    n2, n3, n4, n5, n6 = n * n, n ** 3, n ** 4, n ** 5, n ** 6
    A = (1 + n2 / 4 + n4 / 64 + n6 / 256) / (1 + n)
    alpha = [n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800,
             13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360,
             61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440,
             49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600,
             34729 * n5 / 80640 - 3418889 * n6 / 1995840,
             212378941 * n6 / 319334400]
    beta = [n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800,
            n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720,
            17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720,
            4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600,
            4583 * n5 / 161280 - 108847 * n6 / 3991680,
            20648693 * n6 / 638668800]
    return A, alpha, beta


def tm_forward(lon, lat, lon0: float, a: float = A_GRS80, f: float = F_GRS80, k0: float = K0_UTM, fe: float = 500000.0, fn: float = 0.0):
    """(lon, lat) degrees → (easting, northing) metres of the transverse Mercator projection with central meridian lon0."""
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    n = f / (2 - f)
    e = math.sqrt(f * (2 - f))
    A, alpha, _ = _series(n)
    phi, lam = np.radians(lat), np.radians(lon - lon0)
    s = np.sin(phi)
    tau = np.tan(phi)
    sigma = np.sinh(e * np.arctanh(e * s))
    taup = tau * np.sqrt(1 + sigma * sigma) - sigma * np.sqrt(1 + tau * tau)      # tan of the conformal latitude
    xi = np.arctan2(taup, np.cos(lam))
    eta = np.arcsinh(np.sin(lam) / np.sqrt(taup * taup + np.cos(lam) ** 2))
    x, y = eta.copy(), xi.copy()
    for j, aj in enumerate(alpha, start=1):
        y = y + aj * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
        x = x + aj * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
    return fe + k0 * a * A * x, fn + k0 * a * A * y


def tm_inverse(easting, northing, lon0: float, a: float = A_GRS80, f: float = F_GRS80, k0: float = K0_UTM, fe: float = 500000.0, fn: float = 0.0):
    """(easting, northing) metres → (lon, lat) degrees."""
    easting, northing = np.asarray(easting, dtype=np.float64), np.asarray(northing, dtype=np.float64)
    n = f / (2 - f)
    e = math.sqrt(f * (2 - f))
    A, _, beta = _series(n)
    xi = (northing - fn) / (k0 * a * A)
    eta = (easting - fe) / (k0 * a * A)
    xip, etap = xi.copy(), eta.copy()
    for j, bj in enumerate(beta, start=1):
        xip = xip - bj * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
        etap = etap - bj * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
    taup = np.sin(xip) / np.sqrt(np.sinh(etap) ** 2 + np.cos(xip) ** 2)
    lam = np.arctan2(np.sinh(etap), np.cos(xip))
    tau = taup.copy()                                   # Newton on taup(tau) = taup (Karney eqs. 19 - 21): converges in 2 - 3 steps
    for _ in range(6):
        sigma = np.sinh(e * np.arctanh(e * tau / np.sqrt(1 + tau * tau)))
        tpi = tau * np.sqrt(1 + sigma * sigma) - sigma * np.sqrt(1 + tau * tau)
        dtau = (taup - tpi) / np.sqrt(1 + tpi * tpi) * (1 + (1 - e * e) * tau * tau) / ((1 - e * e) * np.sqrt(1 + tau * tau))
        tau = tau + dtau
    return np.degrees(lam) + lon0, np.degrees(np.arctan(tau))


def _utm(epsg: int) -> Optional[Tuple[float, float]]:
    """(central meridian, false northing) of a UTM code, or None."""
    if 32601 <= epsg <= 32660:
        return (epsg - 32600) * 6 - 183.0, 0.0
    if 32701 <= epsg <= 32760:
        return (epsg - 32700) * 6 - 183.0, 10000000.0
    if 25828 <= epsg <= 25838:
        return (epsg - 25800) * 6 - 183.0, 0.0
    return None


def supported(epsg: int) -> bool:
    return epsg in GEOGRAPHIC or epsg == 3857 or _utm(int(epsg)) is not None


def _to_geographic(xy: np.ndarray, epsg: int) -> np.ndarray:
    if epsg in GEOGRAPHIC:
        return xy
    if epsg == 3857:
        lon = np.degrees(xy[:, 0] / A_GRS80)
        lat = np.degrees(2 * np.arctan(np.exp(xy[:, 1] / A_GRS80)) - math.pi / 2)
        return np.stack([lon, lat], axis=1)
    lon0, fn = _utm(epsg)
    lon, lat = tm_inverse(xy[:, 0], xy[:, 1], lon0, fn=fn)
    return np.stack([lon, lat], axis=1)


def _from_geographic(ll: np.ndarray, epsg: int) -> np.ndarray:
    if epsg in GEOGRAPHIC:
        return ll
    if epsg == 3857:
        x = A_GRS80 * np.radians(ll[:, 0])
        y = A_GRS80 * np.log(np.tan(math.pi / 4 + np.radians(ll[:, 1]) / 2))
        return np.stack([x, y], axis=1)
    lon0, fn = _utm(epsg)
    x, y = tm_forward(ll[:, 0], ll[:, 1], lon0, fn=fn)
    return np.stack([x, y], axis=1)


def transform_points(xy, src_epsg: int, dst_epsg: int) -> np.ndarray:
    """[n,2] coordinates (x = easting / longitude, y = northing / latitude: GIS axis order, as geopandas hands them over)."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    src_epsg, dst_epsg = int(src_epsg), int(dst_epsg)
    if src_epsg == dst_epsg or (src_epsg in GEOGRAPHIC and dst_epsg in GEOGRAPHIC):
        return xy.copy()
    for code in (src_epsg, dst_epsg):
        if not supported(code):
            raise ValueError(f"EPSG:{code} is not among the systems this package reprojects (geographic WGS 84 / ETRS89, their UTM zones, "
                             f"Web Mercator): store the outline in the rasters' CRS (EPSG:{dst_epsg})")
    return _from_geographic(_to_geographic(xy, src_epsg), dst_epsg)


def to_crs(polygons: Sequence[List[np.ndarray]], src_epsg: Optional[int], dst_epsg: Optional[int]) -> List[List[np.ndarray]]:
    """The polygons of a layer (lists of closed rings) in the CRS ``dst_epsg``. A layer or a target without a known code is used
    as it is (a GeoJSON file without a ``crs`` member has always been read in the rasters' CRS by this package)."""
    if not src_epsg or not dst_epsg or int(src_epsg) == int(dst_epsg):
        return [list(p) for p in polygons]
    return [[transform_points(r, src_epsg, dst_epsg) for r in poly] for poly in polygons]
