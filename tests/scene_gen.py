"""Scene generators shared by the tests and by bench.py's f4 configurations (`--config het`, `--config mis`): the fog box of the
reference's own fog render (MitsubaRunner.py:8-40 recipe), and a grid-volume medium in a box."""
import numpy as np


def fog_xml(md="12", rf="gaussian", sensor_medium="", exterior="", env=""):
    return f"""<scene version="3.0.0">
  <integrator type="volpath"><integer name="max_depth" value="{md}"/></integrator>
  <medium type="homogeneous" id="fog"><rgb name="sigma_t" value="1.5, 0.7, 2.0"/><rgb name="albedo" value="0.9, 0.95, 0.6"/>
    <phase type="hg"><float name="g" value="0.5"/></phase></medium>
  <medium type="homogeneous" id="haze"><float name="sigma_t" value="0.05"/><float name="albedo" value="0.8"/>
    <boolean name="has_spectral_extinction" value="false"/></medium>
  <sensor type="perspective"><float name="fov" value="40"/>
    <transform name="to_world"><lookat origin="3, 2.5, 4" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="independent"><integer name="sample_count" value="32"/></sampler>
    <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/><rfilter type="{rf}"/></film>
    {sensor_medium}
  </sensor>
  <shape type="cube"><bsdf type="null"/><ref name="interior" id="fog"/>{exterior}</shape>
  <shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.001"/></transform>
    <bsdf type="diffuse"><texture name="reflectance" type="checkerboard"><transform name="to_uv"><scale x="8" y="8"/></transform></texture></bsdf>{exterior}</shape>
  <shape type="rectangle"><transform name="to_world"><scale value="0.7"/><rotate x="1" angle="90"/><translate y="3.5"/></transform>
    <emitter type="area"><rgb name="radiance" value="20, 18, 15"/></emitter>{exterior}</shape>
  {env}
</scene>"""


def het_xml(vol, sampler="independent", spectral="true", boundary="null", extra_medium="", inside_ref="smoke", md=12):
    return f"""<scene version="3.0.0">
  <integrator type="volpath"><integer name="max_depth" value="{md}"/></integrator>
  <medium type="heterogeneous" id="smoke">
    <volume name="sigma_t" type="gridvolume"><string name="filename" value="{vol}"/>
      <transform name="to_world"><scale value="2"/><translate x="-1" y="-1" z="-1"/></transform></volume>
    <rgb name="albedo" value="0.9, 0.8, 0.6"/><float name="scale" value="3"/><boolean name="has_spectral_extinction" value="{spectral}"/>
    <phase type="hg"><float name="g" value="0.3"/></phase>
  </medium>
  {extra_medium}
  <sensor type="perspective"><float name="fov" value="40"/>
    <transform name="to_world"><lookat origin="3, 2.5, 4" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="{sampler}"><integer name="sample_count" value="16"/></sampler>
    <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/><rfilter type="box"/></film>
  </sensor>
  <shape type="cube"><bsdf type="{boundary}"/><ref name="interior" id="{inside_ref}"/></shape>
  <shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.001"/></transform>
    <bsdf type="diffuse"><texture name="reflectance" type="checkerboard"><transform name="to_uv"><scale x="8" y="8"/></transform></texture></bsdf></shape>
  <shape type="rectangle"><transform name="to_world"><scale value="0.7"/><rotate x="1" angle="90"/><translate y="3.5"/></transform>
    <emitter type="area"><rgb name="radiance" value="20, 18, 15"/></emitter></shape>
  <emitter type="constant"><rgb name="radiance" value="0.3, 0.4, 0.6"/></emitter>
</scene>"""


TWO_MEDIA_FOG = '<medium type="homogeneous" id="fog"><rgb name="sigma_t" value="0.9, 0.3, 1.6"/><rgb name="albedo" value="0.8, 0.8, 0.9"/><boolean name="has_spectral_extinction" value="false"/></medium>'


def two_media_xml(vol, hom=TWO_MEDIA_FOG, **kw):
    """het_xml plus a second box filled with a homogeneous medium next to the grid volume"""
    return het_xml(vol, extra_medium=hom, **kw).replace('<shape type="rectangle"><transform name="to_world"><scale value="6"/>',
        '<shape type="cube"><transform name="to_world"><scale value="0.5"/><translate x="2" y="-0.4"/></transform><bsdf type="null"/><ref name="interior" id="fog"/></shape>'
        '<shape type="rectangle"><transform name="to_world"><scale value="6"/>')


def smoke_grid(seed=3, shape=(12, 10, 8)):
    """the density grid of the heterogeneous-medium tests"""
    return (0.05 + np.random.default_rng(seed).random(shape) ** 3).astype(np.float32)


def resized(xml, width, height, spp):
    """the same scene at another film size / sample count (the generators above hard-code small ones)"""
    import re
    xml = re.sub(r'<integer name="width" value="\d+"/>', f'<integer name="width" value="{width}"/>', xml)
    xml = re.sub(r'<integer name="height" value="\d+"/>', f'<integer name="height" value="{height}"/>', xml)
    return re.sub(r'<integer name="sample_count" value="\d+"/>', f'<integer name="sample_count" value="{spp}"/>', xml)
