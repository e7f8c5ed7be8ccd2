"""treedetection_amd.crs — the ``to_crs`` of the outline layers (reference helpers.py:55,789-790, preprocessing.py:158 →
geopandas → PROJ, absent here). No PROJ to compare with: the Krüger series is checked against facts that do not depend on it —
a numerical quadrature of the meridian arc, conformality of the numerical Jacobian, Snyder's published worked example (Map
Projections: A Working Manual, UTM on the Clarke 1866 ellipsoid), the definition of the UTM grid — and against itself (round trips)."""
import json
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from treedetection_amd import crs  # noqa: E402


def _meridian_arc(lat_deg, a=crs.A_GRS80, f=crs.F_GRS80, n=200001):
    """∫ M(φ) dφ from 0 to lat, M = a (1 - e²) / (1 - e² sin²φ)^(3/2): composite Simpson in float64."""
    e2 = f * (2 - f)
    phi = np.linspace(0.0, math.radians(lat_deg), n)
    m = a * (1 - e2) / (1 - e2 * np.sin(phi) ** 2) ** 1.5
    h = phi[1] - phi[0]
    return h / 3 * (m[0] + m[-1] + 4 * m[1:-1:2].sum() + 2 * m[2:-1:2].sum())


def test_utm_grid_definition_and_meridian_arc():
    # on the central meridian: easting 500 000, northing = 0.9996 x the meridian arc
    for lat in (0.0, 10.0, 47.9, 60.0, 84.0):
        e, n = crs.tm_forward(9.0, lat, 9.0)
        assert abs(float(e) - 500000.0) < 1e-6
        assert abs(float(n) - 0.9996 * _meridian_arc(lat)) < 2e-4, lat            # the quadrature's own accuracy
    # southern hemisphere: false northing 10 000 km
    assert abs(crs.transform_points([[9.0, -10.0]], 4326, 32732)[0, 1] - (10000000.0 - 0.9996 * _meridian_arc(10.0))) < 2e-4
    # zones: EPSG:25832 = ETRS89 / UTM 32N = central meridian 9 E; 32633 = 15 E
    assert crs._utm(25832) == (9.0, 0.0) and crs._utm(32633) == (15.0, 0.0) and crs._utm(32719) == (-69.0, 10000000.0)


def test_snyders_worked_example_clarke_1866():
    """Snyder (1987), numerical example for the UTM: Clarke 1866 ellipsoid, lat 40 30' N, lon 73 30' W, central meridian 75 W,
    k0 = 0.9996 → x = 127 106.5 m, y = 4 484 124.4 m."""
    a, f = 6378206.4, 1.0 - math.sqrt(1.0 - 0.00676866)
    x, y = crs.tm_forward(-73.5, 40.5, -75.0, a=a, f=f, fe=0.0)
    assert abs(float(x) - 127106.5) < 0.1 and abs(float(y) - 4484124.4) < 0.1
    lon, lat = crs.tm_inverse(127106.5, 4484124.4, -75.0, a=a, f=f, fe=0.0)
    assert abs(float(lon) + 73.5) < 2e-6 and abs(float(lat) - 40.5) < 2e-6


def test_projection_is_conformal_and_round_trips():
    rng = np.random.default_rng(0)
    lon = rng.uniform(5.5, 12.5, 2000)              # up to 3.5 degrees off the central meridian
    lat = rng.uniform(-80, 84, 2000)
    e, n = crs.tm_forward(lon, lat, 9.0)
    lon2, lat2 = crs.tm_inverse(e, n, 9.0)
    assert np.abs(lon2 - lon).max() < 1e-11 and np.abs(lat2 - lat).max() < 1e-11          # < 1e-6 m
    # conformal: the Jacobian with respect to metres east / north on the ellipsoid is a rotation times a scale
    d = 1e-6
    e2 = crs.F_GRS80 * (2 - crs.F_GRS80)
    for lo, la in ((10.7, 48.1), (6.2, 51.0), (11.9, -33.0)):
        phi = math.radians(la)
        nu = crs.A_GRS80 / math.sqrt(1 - e2 * math.sin(phi) ** 2)
        rho = crs.A_GRS80 * (1 - e2) / (1 - e2 * math.sin(phi) ** 2) ** 1.5
        ex, nx = (np.array(crs.tm_forward(lo + d, la, 9.0)) - np.array(crs.tm_forward(lo - d, la, 9.0))) / (2 * math.radians(d) * nu * math.cos(phi))
        ey, ny = (np.array(crs.tm_forward(lo, la + d, 9.0)) - np.array(crs.tm_forward(lo, la - d, 9.0))) / (2 * math.radians(d) * rho)
        assert abs(ex - ny) < 1e-6 and abs(ey + nx) < 1e-6                                 # Cauchy-Riemann
        k = math.hypot(ex, nx)
        assert 0.9996 <= k < 1.0010                                                         # the UTM scale factor inside a zone


def test_transform_points_between_codes():
    pts = np.array([[11.5761, 48.1372], [9.0, 0.0], [8.4037, 49.0069]])
    utm = crs.transform_points(pts, 4326, 25832)
    back = crs.transform_points(utm, 25832, 4258)
    assert np.abs(back - pts).max() < 1e-10
    assert abs(utm[1, 0] - 500000.0) < 1e-6 and abs(utm[1, 1]) < 1e-6
    # Munich's Marienplatz lies ~ 692 km east of the false origin's 500 km... in zone 32: E ~ 691.6 km, N ~ 5334.8 km
    assert 691000 < utm[0, 0] < 692500 and 5334000 < utm[0, 1] < 5336000
    # neighbouring zones through the geographic system; Web Mercator and back
    z33 = crs.transform_points(utm, 25832, 32633)
    assert np.abs(crs.transform_points(z33, 32633, 25832) - utm).max() < 1e-5
    wm = crs.transform_points(pts, 4326, 3857)
    assert abs(wm[0, 0] - crs.A_GRS80 * math.radians(11.5761)) < 1e-6
    assert np.abs(crs.transform_points(wm, 3857, 4326) - pts).max() < 1e-10
    assert crs.transform_points(pts, 4326, 4258) is not pts and np.array_equal(crs.transform_points(pts, 4326, 4258), pts)
    with pytest.raises(ValueError, match="EPSG:31467"):
        crs.transform_points(pts, 31467, 25832)                                            # DHDN / Gauss-Krüger: another datum, refused
    assert crs.to_crs([[pts]], None, 25832)[0][0] is pts                                   # unknown source: used as it is


def test_outline_in_geographic_coordinates_flags_the_same_tiles(tmp_path):
    """preprocessing.tile_data with a forest outline stored in EPSG:4326 (what GeoJSON files are by default) against the same
    outline stored in the rasters' EPSG:25832: identical only_forest / only_urban flags (reference preprocessing.py:157-158 to_crs)."""
    from treedetection_amd.geotiff import write_geotiff
    from treedetection_amd.preprocessing import tile_data
    x0, y1 = 412000.0, 5318200.0
    write_geotiff(str(tmp_path / "1.tif"), np.zeros((3, 1000, 1500), np.uint8), (0.2, 0.0, x0, 0.0, -0.2, y1), 25832)
    ring = [[x0 - 50, y1 - 250], [x0 + 137.3, y1 - 250], [x0 + 137.3, y1 + 50], [x0 - 50, y1 + 50], [x0 - 50, y1 - 250]]
    flags = {}
    for name, coords, code in (("utm", ring, 25832), ("geo", crs.transform_points(ring, 25832, 4326).tolist(), 4326)):
        gj = {"type": "FeatureCollection", "crs": {"type": "name", "properties": {"name": f"urn:ogc:def:crs:EPSG::{code}"}},
              "features": [{"type": "Feature", "properties": {}, "geometry": {"type": "Polygon", "coordinates": [coords]}}]}
        (tmp_path / f"{name}.geojson").write_text(json.dumps(gj))
        tile_data([str(tmp_path / "1.tif")], str(tmp_path / f"tiles_{name}"), buffer=5, tile_width=50, tile_height=50,
                  forest_shapefile=str(tmp_path / f"{name}.geojson"))
        meta = json.load(open(tmp_path / f"tiles_{name}" / "1.json"))
        flags[name] = {k: (v["only_forest"], v["only_urban"]) for k, v in meta.items()}
    assert flags["utm"] == flags["geo"]
    assert any(f for f, _ in flags["utm"].values()) and any(u for _, u in flags["utm"].values())
