-- A tilted stereographic lens: the projection comes from a helper module shared with other lenses, the tilt is a rotation matrix.
local optics = require "examples.lenses.shared.optics"
local c, s = optics.tilt(20)

max_fov = 300
max_vfov = 300
lens_width = 4
lens_height = 4
onload = "f_contain"

function lens_inverse(x, y)
   local r = sqrt(x * x + y * y)
   if r == 0 then return 0, s, c end
   local theta = optics.stereographic_angle(r)
   local k = sin(theta) / r
   local v = {x * k, y * k, cos(theta)}
   local rot = {{1, 0, 0}, {0, c, s}, {0, -s, c}}          -- about the x axis
   local out = {0, 0, 0}
   for i = 1, #rot do
      for j = 1, #rot[i] do out[i] = out[i] + rot[i][j] * v[j] end
   end
   return out[1], out[2], out[3]
end
