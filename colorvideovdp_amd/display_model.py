"""Display models (host side), mirroring the constructors and getters of
pycvvdp/display_model.py: vvdp_display_photometry.load (:156-200), vvdp_display_photo_eotf (:301-376),
vvdp_display_geometry (:441-526, :588-626).

Only parameters live here; the per-pixel forward model (EOTF, black level, reflections, RGB->DKL) runs
in the HIP photometry kernel (csrc/photometry.hip).
"""
import logging
import math

import numpy as np

from .config import load_config

# display_model.py:17-25
XYZ_to_LMS2006 = ((0.187596268556126, 0.585168649077728, -0.026384263306304),
                  (-0.133397430663221, 0.405505777260049, 0.034502127690364),
                  (0.000244379021663, -0.000542995890619, 0.019406849066323))
LMS2006_to_DKLd65 = ((1.000000000000000, 1.000000000000000, 0),
                     (1.000000000000000, -2.311130179947035, 0),
                     (-1.000000000000000, -1.000000000000000, 50.977571328718781))

EOTF_IDS = {"sRGB": 0, "PQ": 1, "HLG": 2, "linear": 3}


class vvdp_display_photometry:
    def __init__(self, source_colorspace="sRGB", config_paths=[]):
        spaces = load_config("color_spaces.json", config_paths)
        if source_colorspace not in spaces:
            raise RuntimeError(f'Color space: "{source_colorspace}" not found')
        cs = spaces[source_colorspace]
        if "RGB2X" in cs:
            self.rgb2xyz_list = [cs["RGB2X"], cs["RGB2Y"], cs["RGB2Z"]]
        self.EOTF = cs["EOTF"]

    @classmethod
    def list_displays(cls, config_paths):
        for name in load_config("display_models.json", config_paths):
            cls.load(name, config_paths).print()

    @classmethod
    def load(cls, display_name, config_paths):
        models = load_config("display_models.json", config_paths)
        if display_name not in models:
            logging.error(f"Display model: '{display_name}' not found")
            raise RuntimeError("Display model not found")
        m = models[display_name]
        Y_peak = m["max_luminance"]
        if "min_luminance" in m:
            contrast = Y_peak / m["min_luminance"]
        else:
            contrast = m.get("contrast", 500)
        obj = vvdp_display_photo_eotf(Y_peak, contrast=contrast, source_colorspace=m.get("colorspace", "sRGB"),
                                      E_ambient=m.get("E_ambient", 0), k_refl=m.get("k_refl", 0.005), name=display_name,
                                      exposure=m.get("exposure", 1), config_paths=config_paths)
        obj.full_name = m["name"]
        obj.short_name = display_name
        return obj


class vvdp_display_photo_eotf(vvdp_display_photometry):
    def __init__(self, Y_peak, contrast=1000, source_colorspace="sRGB", EOTF=None, E_ambient=0, k_refl=0.005, exposure=1,
                 name=None, config_paths=[]):
        super().__init__(source_colorspace=source_colorspace, config_paths=config_paths)
        if EOTF is not None:
            self.EOTF = EOTF
        self.Y_peak = Y_peak
        self.contrast = contrast
        self.E_ambient = E_ambient
        self.k_refl = k_refl
        self.name = name
        self.exposure = exposure

    def is_input_display_encoded(self):
        return self.EOTF != "linear"

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return NotImplemented
        return (self.Y_peak, self.contrast, self.EOTF, self.E_ambient, self.k_refl, self.exposure) == \
               (other.Y_peak, other.contrast, other.EOTF, other.E_ambient, other.k_refl, other.exposure)

    def get_peak_luminance(self):
        return self.Y_peak

    def get_black_level(self):
        Y_refl = self.E_ambient / math.pi * self.k_refl
        Y_black = self.Y_peak / self.contrast
        return Y_black, Y_refl

    def print(self):
        Y_black, Y_refl = self.get_black_level()
        logging.info("Photometric display model: {}".format(self.name))
        logging.info("  Peak luminance: {} cd/m^2".format(self.Y_peak))
        logging.info("  EOTF: {}".format(self.EOTF))
        logging.info("  Contrast - theoretical: {}:1".format(round(self.contrast)))
        logging.info("  Contrast - effective: {}:1".format(round(self.Y_peak / (Y_black + Y_refl))))
        logging.info("  Ambient light: {} lux".format(self.E_ambient))
        logging.info("  Display reflectivity: {}%".format(self.k_refl * 100))

    # ---- parameters handed to the HIP core --------------------------------------------------
    def rgb2dkl_fp32(self):
        """fp32 left-to-right product of display_model.py:255-256."""
        if not hasattr(self, "rgb2xyz_list"):
            return np.eye(3, dtype=np.float32)  # 'luminance' colour space: 1-channel content only
        r = np.asarray(self.rgb2xyz_list, dtype=np.float32)
        a = np.asarray(LMS2006_to_DKLd65, dtype=np.float32)
        b = np.asarray(XYZ_to_LMS2006, dtype=np.float32)
        return ((a @ b).astype(np.float32) @ r).astype(np.float32)

    def eotf_params(self):
        """(eotf id, gamma) for the kernel; gamma also carries the HLG system gamma (display_model.py:350-355)."""
        e = self.EOTF
        if e in EOTF_IDS:
            gamma = 0.0
            if e == "HLG":
                gamma = 1.2
                if self.Y_peak > 1000:
                    gamma = 1.2 + 0.42 * math.log10(self.Y_peak / 1000) - 0.07623 * math.log10(self.E_ambient / 5)
            return EOTF_IDS[e], gamma
        if e[0].isnumeric():
            return 4, float(e)
        raise RuntimeError(f"Unknown EOTF '{e}'")


class vvdp_display_geometry:
    def __init__(self, resolution, distance_m=None, distance_display_heights=None, fov_horizontal=None, fov_vertical=None,
                 fov_diagonal=None, diagonal_size_inches=None, ppd=None):
        self.resolution = resolution
        ar = resolution[0] / resolution[1]
        if ppd is not None:
            self.fixed_ppd = ppd
            return
        self.fixed_ppd = None
        if diagonal_size_inches is not None:
            height_mm = math.sqrt((diagonal_size_inches * 25.4) ** 2 / (1 + ar ** 2))
            self.display_size_m = (ar * height_mm / 1000, height_mm / 1000)
        if distance_m is not None and distance_display_heights is not None:
            raise RuntimeError("You can pass only one of: distance_m, distance_display_heights.")
        if distance_m is not None:
            self.distance_m = distance_m
        elif distance_display_heights is not None:
            if not hasattr(self, "display_size_m"):
                raise RuntimeError("You need to specify display diagonal size diagonal_size_inches to specify viewing distance as distance_display_heights")
            self.distance_m = distance_display_heights * self.display_size_m[1]
        elif fov_horizontal is not None or fov_vertical is not None or fov_diagonal is not None:
            self.distance_m = 3
        else:
            raise RuntimeError("Viewing distance must be specified as distance_m or distance_display_heights.")
        if (fov_horizontal is not None) + (fov_vertical is not None) + (fov_diagonal is not None) > 1:
            raise RuntimeError("You can pass only one of fov_horizontal, fov_vertical, fov_diagonal.")
        if fov_horizontal is not None:
            width_m = 2 * math.tan(math.radians(fov_horizontal / 2)) * self.distance_m
            self.display_size_m = (width_m, width_m / ar)
        elif fov_vertical is not None:
            height_m = 2 * math.tan(math.radians(fov_vertical / 2)) * self.distance_m
            self.display_size_m = (height_m * ar, height_m)
        elif fov_diagonal is not None:
            distance_px = math.sqrt(resolution[0] ** 2 + resolution[1] ** 2) / (2.0 * math.tan(math.radians(fov_diagonal * 0.5)))
            height_deg = math.degrees(math.atan(resolution[1] / 2 / distance_px)) * 2
            height_m = 2 * math.tan(math.radians(height_deg / 2)) * self.distance_m
            self.display_size_m = (height_m * ar, height_m)
        self.display_size_deg = (2 * math.degrees(math.atan(self.display_size_m[0] / (2 * self.distance_m))),
                                 2 * math.degrees(math.atan(self.display_size_m[1] / (2 * self.distance_m))))

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return NotImplemented
        return self.__dict__ == other.__dict__

    def get_ppd(self, eccentricity=None):
        if self.fixed_ppd is not None:
            return self.fixed_ppd
        pix_deg = 2 * math.degrees(math.atan(0.5 * self.display_size_m[0] / self.resolution[0] / self.distance_m))
        base_ppd = 1 / pix_deg
        if eccentricity is None:
            return base_ppd
        delta = pix_deg / 2
        tan_delta = math.tan(math.radians(delta))
        ecc = np.asarray(eccentricity, dtype=np.float64)
        return base_ppd * (np.tan(np.radians(ecc + delta)) - np.tan(np.radians(ecc))) / tan_delta

    def print(self):
        logging.info("Geometric display model:")
        if self.fixed_ppd is not None:
            logging.info("  Fixed pixels-per-degree: {}".format(self.fixed_ppd))
        else:
            logging.info("  Resolution: {w} x {h} pixels".format(w=self.resolution[0], h=self.resolution[1]))
            logging.info("  Display size: {w:.1f} x {h:.1f} cm".format(w=self.display_size_m[0] * 100, h=self.display_size_m[1] * 100))
            logging.info("  Viewing distance: {d:.3f} m".format(d=self.distance_m))
            logging.info("  Pixels-per-degree (center): {ppd:.2f}".format(ppd=self.get_ppd()))

    @classmethod
    def load(cls, display_name, config_paths=[]):
        models = load_config("display_models.json", config_paths)
        if display_name not in models:
            logging.error(f"Display model: '{display_name}' not found")
            raise RuntimeError("Display model not found")
        m = models[display_name]
        assert "resolution" in m
        W, H = m["resolution"]
        if "pixels_per_degree" in m:
            return cls((W, H), ppd=m["pixels_per_degree"])
        if "viewing_distance_meters" in m:
            distance_m = m["viewing_distance_meters"]
        elif "viewing_distance_inches" in m:
            distance_m = m["viewing_distance_inches"] * 0.0254
        else:
            distance_m = None
        if "diagonal_size_meters" in m:
            diag = m["diagonal_size_meters"] / 0.0254
        else:
            diag = m.get("diagonal_size_inches")
        return cls((W, H), distance_m=distance_m, fov_diagonal=m.get("fov_diagonal"), diagonal_size_inches=diag)
